// Closed-loop Pipeline.step on the device (SURVEY.md 8f "next" row 3, the State half): the per-frame state of the reference --
// State(landmarks, landmarks_kp, candidates_kp, trajectory) and the dead lists (src/state/state.py:4-10, src/pipeline/pipeline.py:31) --
// as device tables, and the glue of Pipeline.step (pipeline.py:92-167) between the stage kernels as five list kernels:
//
//   k_pipe_extend     after KLT     Extractor.extend_tracks / extend_landmarks (extractor.py:38-88) + pipeline.py:101-102
//   k_pipe_prune      after PnP     pipeline.py:124-140 (inlier pruning, trajectory.append) + the split of triangulate_tracks (extractor.py:202-203)
//   k_pipe_promote    after DLT     triangulate.py:87-111 filters, extractor.py:231-240 gate, pipeline.py:153-154; then the selection half of
//                                   BundleAdjuster.adjust (bundle_adjuster.py:132-176): resurrection, observation table, x0
//                     also: the resident point set (exclusion discs, next KLT); then S-T + spawn on a side stream beside the BA
//   (k_pipe_dense     the resident point set alone: after a host write of the tables, or when promote is not among the stages)
//   k_pipe_writeback  after BA      bundle_adjuster.py:197-213
//   k_pipe_spawn      after S-T     extractor.py:127-131 / pipeline.py:159-163; free rows rebuilt; the frame's record
//
// The reference relies on OBJECT IDENTITY: adjust appends recently dead landmarks to state._landmarks without copying
// (bundle_adjuster.py:142-147), extend_landmarks deep-copies the keypoint of a survivor but keeps the landmark object
// (extractor.py:80-86), Pipeline.step deep-copies what dies (one deepcopy call per list: entries sharing a landmark object share
// the copy).  So the tables are OBJECT ROWS with stable indices (K = Keypoint, L = Landmark) and the lists hold row indices; several
// entries may refer to one row.  oracle/pipe_oracle.py is the same algorithm in numpy, checked object by object against the
// reference's loop over Python objects; this file follows it phase by phase.
//
// One workgroup (1024 threads) per sequence; a thread owns CH = ceil(max_pts / 1024) <= 4 CONSECUTIVE list entries, reads them all
// before anything is written, and ordered compaction / allocation are block-wide prefix scans, so results do not depend on timing.
// What such a kernel costs is its number of DEPENDENT trips to global memory: extend, prune and promote read the lists, then every row field
// and the heads of the free lists in one trip each, and keep their bookkeeping (elections, counts, the growing list) in LDS.
#include "vo_internal.h"

#include <math.h>
#include <string.h>

#define PIPE_TPB 1024
#define PIPE_CH 8                 // most list entries per thread: max_pts <= PIPE_TPB * PIPE_CH = 8 192
#define PIPE_HIST VO_PIPE_HIST
#define PIPE_NCNT 32

// counters [B][PIPE_NCNT]
enum { C_NCAND = 0, C_NLM = 1, C_NDEAD = 2, C_NINERT = 3, C_STATUS = 4, C_T = 5,
       C_NFREEK = 6, C_NFREEL = 7, C_HEADK = 8, C_HEADL = 9, C_NRIPE = 10, C_NPTS = 11, C_NNEW = 12, C_NRES = 13, C_NDET = 14,
       C_OVERFLOW = 15, C_NOBS = 16, C_NKLT = 17, C_NPNP = 19 };

struct pipe_ptrs {
  int32_t *k_tf, *k_tt, *k_len; float2 *k_uv, *k_first, *k_hist;     // K rows [B][R]; hist [B][HIST][R]
  int32_t* l_tl; double* l_p;                                         // L rows [B][R], [B][R][3]
  int32_t *cand, *lm_L, *lm_K, *lm_ksh, *dead_L, *dead_K, *ripe;      // lists [B][N]
  int32_t* cnt;                                                       // [B][PIPE_NCNT]
  int32_t *freeK, *freeL, *scr;                                       // [B][R]; scr: one word per landmark row for k_pipe_writeback's election (the other
                                                                      // list kernels keep theirs in LDS)
  double* H;                                                          // [B][HIST][12]
  double* Kc;                                                         // [B][9]
  int32_t* dn;                                                        // [4][B] dense per-sequence counts the stage kernels index by sequence:
                                                                      // resident points (KLT, exclusion discs) | PnP correspondences | DLT pairs
  int N, R;
};
enum { DN_PTS = 0, DN_PNP = 1, DN_RIPE = 2, DN_ROOM = 3 };   // DN_ROOM: free slots of the table + 1 = the corners the frame's re-detection can use

struct vo_pipe_ws {
  int N = 0, R = 0;
  vo_pipe_params prm;
  void* tab[VO_PIPE_N_TABLES] = {};              // all inside ONE allocation (tab_slab), in table order: the lists, the counters and the
  size_t tab_bytes[VO_PIPE_N_TABLES] = {};       // trajectory ring (tables VO_PIPE_CAND ...) are contiguous -> vo_pipe_lists_read is one copy.  (per sequence)
  uint8_t* tab_slab = nullptr;
  size_t lists_bytes = 0;                        // from tab[VO_PIPE_CAND] to the end of the slab
  uint8_t* d_gather = nullptr;                   // vo_pipe_rows_read: packed rows [N][288 B]
  int32_t* d_gather_rows = nullptr;
  int32_t *d_ripe = nullptr, *d_freeK = nullptr, *d_freeL = nullptr, *d_scr = nullptr, *d_dn = nullptr, *d_cam_sel = nullptr;
  vo_dlt_cam* d_cams = nullptr;                  // [B][HIST]: one camera pair per birth frame of the ripe candidates
  double* d_K = nullptr;
  vo_pipe_record* d_rec = nullptr;               // [B]
  vo_pipe_record* h_rec = nullptr;               // pinned [VO_PIPE_INFLIGHT][B]
  hipEvent_t ev[VO_PIPE_INFLIGHT] = {};
  hipEvent_t ev_track = nullptr;                 // the side stream's pyramid + KLT of a step are done
  long enq = 0, fetched = 0;
  bool lm_half_pending = false;                  // a TRACK | TRACK_CANDIDATES call has tracked every keypoint and applied the candidates' half only:
                                                 // the landmarks' half (TRACK_LANDMARKS alone) may follow -- and only then
};

static pipe_ptrs pipe_make(const vo_pipe_ws* w) {
  pipe_ptrs P;
  P.k_tf = (int32_t*)w->tab[VO_PIPE_K_TFIRST]; P.k_tt = (int32_t*)w->tab[VO_PIPE_K_TTOTAL]; P.k_len = (int32_t*)w->tab[VO_PIPE_K_HISTLEN];
  P.k_uv = (float2*)w->tab[VO_PIPE_K_UV]; P.k_first = (float2*)w->tab[VO_PIPE_K_UVFIRST]; P.k_hist = (float2*)w->tab[VO_PIPE_K_HIST];
  P.l_tl = (int32_t*)w->tab[VO_PIPE_L_TLATEST]; P.l_p = (double*)w->tab[VO_PIPE_L_P];
  P.cand = (int32_t*)w->tab[VO_PIPE_CAND]; P.lm_L = (int32_t*)w->tab[VO_PIPE_LM_L]; P.lm_K = (int32_t*)w->tab[VO_PIPE_LM_K];
  P.lm_ksh = (int32_t*)w->tab[VO_PIPE_LM_KSHARED]; P.dead_L = (int32_t*)w->tab[VO_PIPE_DEAD_L]; P.dead_K = (int32_t*)w->tab[VO_PIPE_DEAD_K];
  P.ripe = w->d_ripe; P.cnt = (int32_t*)w->tab[VO_PIPE_COUNTS]; P.freeK = w->d_freeK; P.freeL = w->d_freeL; P.scr = w->d_scr;
  P.H = (double*)w->tab[VO_PIPE_POSES]; P.Kc = w->d_K; P.dn = w->d_dn; P.N = w->N; P.R = w->R;
  return P;
}

// tables of sequence b
__device__ __forceinline__ pipe_ptrs pipe_select(pipe_ptrs P, int b) {
  const size_t r = (size_t)b * P.R, n = (size_t)b * P.N;
  P.k_tf += r; P.k_tt += r; P.k_len += r; P.k_uv += r; P.k_first += r; P.k_hist += r * PIPE_HIST;
  P.l_tl += r; P.l_p += 3 * r;
  P.cand += n; P.lm_L += n; P.lm_K += n; P.lm_ksh += n; P.dead_L += n; P.dead_K += n; P.ripe += n;
  P.cnt += (size_t)b * PIPE_NCNT; P.freeK += r; P.freeL += r; P.scr += r;
  P.H += (size_t)b * PIPE_HIST * 12; P.Kc += 9 * (size_t)b;
  return P;
}

// values other threads of the workgroup update with atomics are read / written through the same path (the L2), not the vector L1
__device__ __forceinline__ int ld_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_i32(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// block-wide exclusive scan of a 0/1 flag over the 1024 threads in thread order; s_w: 16 ints of LDS
__device__ __forceinline__ int pipe_scan(int flag, int* s_w, int& total) {
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

// A thread owns the CH CONSECUTIVE list entries j = tid * CH + c, so list order is thread order and ONE block-wide scan of the
// per-thread counts ranks the flagged entries: rank[c] = position of entry (tid, c) among the flagged ones; returns their number.
template <int CH>
__device__ __forceinline__ int pipe_rank(const bool* flag, int* rank, int* s_w) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int cnt = 0;
#pragma unroll
  for (int c = 0; c < CH; c++) cnt += flag[c] ? 1 : 0;
  int x = cnt;                           // inclusive scan inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
  __syncthreads();                       // s_w may still be read from the previous call
  if (lane == 63) s_w[wave] = x;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) { const int v = s_w[w]; if (w < wave) off += v; tot += v; }
  int r = off + x - cnt;
#pragma unroll
  for (int c = 0; c < CH; c++) { rank[c] = r; r += flag[c] ? 1 : 0; }
  return tot;
}

// block-wide exclusive scan of one count per thread, in thread order; returns the offset of this thread, total = sum over the workgroup
__device__ __forceinline__ int pipe_scan_count(int cnt, int* s_w, int& total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int x = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
  __syncthreads();                       // s_w may still be read from the previous call
  if (lane == 63) s_w[wave] = x;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) { const int v = s_w[w]; if (w < wave) off += v; tot += v; }
  total = tot;
  return off + x - cnt;
}

__device__ __forceinline__ float2* pipe_hist_slot(const pipe_ptrs& P, int idx) { return P.k_hist + (size_t)(idx & (PIPE_HIST - 1)) * P.R; }

__device__ __forceinline__ void pipe_copy_K(const pipe_ptrs& P, int dst, int src) {
  P.k_tf[dst] = P.k_tf[src]; P.k_tt[dst] = P.k_tt[src]; P.k_len[dst] = P.k_len[src];
  P.k_uv[dst] = P.k_uv[src]; P.k_first[dst] = P.k_first[src];
#pragma unroll 8
  for (int h = 0; h < PIPE_HIST; h++) P.k_hist[(size_t)h * P.R + dst] = P.k_hist[(size_t)h * P.R + src];
}

// (free rows are handed out from the front of the ascending free lists: head counters C_HEADK / C_HEADL, rebuilt by k_pipe_spawn)
__device__ __forceinline__ bool pipe_inside(float2 q, int W, int H) {
  return q.x >= 0.f && q.x <= (float)W && q.y >= 0.f && q.y <= (float)H;       // ends included; NaN fails (extractor.py:53,75)
}

// vec -> R (so3.rodrigues_vec_to_mat / cv2.Rodrigues)
__device__ inline void pipe_rodrigues(const double* r, double* R) {
  const double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (th < 2.220446049250313e-16) { for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
  const double kx = r[0] / th, ky = r[1] / th, kz = r[2] / th, c = cos(th), s = sin(th), c1 = 1.0 - c;
  R[0] = c + c1 * kx * kx;      R[1] = c1 * kx * ky - s * kz; R[2] = c1 * kx * kz + s * ky;
  R[3] = c1 * ky * kx + s * kz; R[4] = c + c1 * ky * ky;      R[5] = c1 * ky * kz - s * kx;
  R[6] = c1 * kz * kx - s * ky; R[7] = c1 * kz * ky + s * kx; R[8] = c + c1 * kz * kz;
}

// R -> vec (so3.rodrigues_mat_to_vec without the SVD projection: R is a rotation to rounding error)
__device__ inline void pipe_log_so3(const double* R, double* r) {
  const double v[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  const double s = sqrt(0.25 * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
  double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
  c = fmin(1.0, fmax(-1.0, c));
  const double th = acos(c);
  if (s < 1e-5) {
    if (c > 0) { r[0] = r[1] = r[2] = 0.0; return; }
    double rx = sqrt(fmax((R[0] + 1) * 0.5, 0.0));
    double ry = sqrt(fmax((R[4] + 1) * 0.5, 0.0)) * (R[1] < 0 ? -1.0 : 1.0);
    double rz = sqrt(fmax((R[8] + 1) * 0.5, 0.0)) * (R[2] < 0 ? -1.0 : 1.0);
    if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && ((R[5] > 0) != (ry * rz > 0))) rz = -rz;
    const double n = sqrt(rx * rx + ry * ry + rz * rz);
    const double k = n > 0 ? th / n : 0.0;
    r[0] = rx * k; r[1] = ry * k; r[2] = rz * k;
    return;
  }
  const double k = 0.5 * th / s;
  r[0] = v[0] * k; r[1] = v[1] * k; r[2] = v[2] * k;
}

// ================================================================================================
// k_pipe_extend: the keep rule and bookkeeping after the KLT of [landmark keypoints | candidates]
// ================================================================================================
// The kernel is a chain of phases of ONE workgroup, so what it costs is the number of dependent trips to global memory, not the work: the
// first form made ~24 of them (row fields fetched where they were needed, free rows fetched when their rank was known, the leader election
// and the t_latest increments as global atomics, every row copy a chain of 32 load-store pairs in one thread: 31 us for one sequence,
// 75 in a batch of 32).  Now: (1) counters, (2) the lists and the tracked points, (3) EVERY row field any phase will need plus a window
// of both free lists into LDS, then all bookkeeping on registers and LDS words (one word per landmark row: survivor count, then the
// election), (4) the row copies as one cooperative pass of the whole workgroup over a work list.
// Dynamic LDS: int32 [R] per-landmark-row word | [N] free K rows | [N] free L rows | [N] copy sources  (28 bytes per slot of the table)
template <int CH>
// which: bit 0 the candidates (extend_tracks), bit 1 the landmarks (extend_landmarks) -- both in the closed loop; the object boundary
// (vo_mi355x/lazy.py) runs them as the reference calls them, one after the other on the SAME tracked point set (the step counter advances
// with the landmarks)
__global__ void __launch_bounds__(PIPE_TPB) k_pipe_extend(pipe_ptrs Pall, const float* __restrict__ pts, size_t slab_seq, int W, int H,
                                                          float* __restrict__ pnp_X, float* __restrict__ pnp_uv, int pnp_cap, int which,
                                                          uint8_t* __restrict__ keep_out) {
  extern __shared__ int32_t s_dyn[];
  __shared__ int s_w[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const pipe_ptrs P = pipe_select(Pall, b);
  if (P.cnt[C_STATUS]) return;
  // one word per landmark ROW (survivor count, then the election): in LDS up to 4 096 slots; above that (CH = 8: 16 384 + rows would not fit
  // beside the three windows) in the global scratch words of the sequence, through the L2 like every value the workgroup's atomics touch
  constexpr bool ROWS_GLOBAL = CH > 4;
  int32_t* const s_row = ROWS_GLOBAL ? P.scr : s_dyn;
  int32_t* const s_fk = ROWS_GLOBAL ? s_dyn : s_dyn + P.R;
  auto row_st = [&](int r, int v) { if (ROWS_GLOBAL) st_i32(&s_row[r], v); else s_row[r] = v; };
  auto row_ld = [&](int r) { return ROWS_GLOBAL ? ld_i32(&s_row[r]) : s_row[r]; };
  int32_t* const s_fl = s_fk + P.N;
  int32_t* const s_src = s_fl + P.N;
  const float2* p1 = reinterpret_cast<const float2*>(vo_seq(pts, slab_seq, b));
  pnp_X += (size_t)b * pnp_cap * 3; pnp_uv += (size_t)b * pnp_cap * 2;
  // ---- trip 1: counters ----
  const int nl = P.cnt[C_NLM], nc = P.cnt[C_NCAND], nd0 = P.cnt[C_NDEAD];
  int headK = P.cnt[C_HEADK], headL = P.cnt[C_HEADL];
  const int nfK = P.cnt[C_NFREEK], nfL = P.cnt[C_NFREEL];
  int overflow = 0, inert = 0;
  __syncthreads();                                   // every thread has read the counters before thread 0 rewrites them

  // ---- trip 2: the lists and the tracked positions ----
  // LATE (CH >= 4): the landmark positions -- 6 CH registers that only the copies of dead landmarks and the PnP input at the very end need -- are
  // fetched there instead.  SPLIT (CH = 8): the candidates' half also fetches its own fields and finishes before the landmarks' row fields are
  // requested (one dependent trip more, half the registers at the peak).  With everything prefetched the CH = 8 kernel kept 380 bytes per
  // lane in scratch (a lane of a 1 024-lane workgroup has 128 registers) and took 123 us in a batch of 32.
  constexpr bool LATE = CH >= 4, SPLIT = CH >= 8;
  constexpr int CE = LATE ? 1 : CH, CS = SPLIT ? 1 : CH;
  int L[CH], K[CH], KC[CS];
  bool keep[CH], die[CH], ksh[CH], keepc[CS];
  float2 q[CH], qc[CS];
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const int j = tid * CH + c;
    L[c] = 0; K[c] = 0; keep[c] = die[c] = ksh[c] = false;
    q[c] = make_float2(0.f, 0.f);
    if ((which & 2) && j < nl) {
      L[c] = P.lm_L[j]; K[c] = P.lm_K[j]; ksh[c] = P.lm_ksh[j] != 0; q[c] = p1[j];
      keep[c] = pipe_inside(q[c], W, H); die[c] = !keep[c];
      keep_out[(size_t)b * pnp_cap + j] = keep[c] ? 1 : 0;      // which landmark entries survived (vo_pipe_inliers_read until the POSE stage overwrites it)
    }
    if (!SPLIT) {
      KC[c] = 0; keepc[c] = false; qc[c] = make_float2(0.f, 0.f);
      if ((which & 1) && j < nc) { KC[c] = P.cand[j]; qc[c] = p1[nl + j]; keepc[c] = pipe_inside(qc[c], W, H); }
    }
  }
  // ---- trip 3: every row field the phases below need, and the heads of the free lists ----
  int len[CH], tt[CH], tl[CH], lenc[CS], ttc[CS];
  double lp[CE][3];
  auto lm_fields = [&]() {
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const int j = tid * CH + c;
      len[c] = tt[c] = tl[c] = 0;
      if (!LATE) lp[c][0] = lp[c][1] = lp[c][2] = 0.0;
      if ((which & 2) && j < nl) {
        len[c] = P.k_len[K[c]]; tt[c] = P.k_tt[K[c]]; tl[c] = P.l_tl[L[c]];
        if (!LATE) for (int k = 0; k < 3; k++) lp[c][k] = P.l_p[3 * (size_t)L[c] + k];
        row_st(L[c], 0);                             // (entries that share a row all write 0)
      }
    }
  };
  if (!SPLIT) {
    lm_fields();
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const int j = tid * CH + c;
      lenc[c] = ttc[c] = 0;
      if ((which & 1) && j < nc) { lenc[c] = P.k_len[KC[c]]; ttc[c] = P.k_tt[KC[c]]; }
    }
  }
  for (int i = tid; i < P.N; i += PIPE_TPB) {        // no phase takes more than N rows of either kind
    s_fk[i] = (headK + i < nfK) ? P.freeK[headK + i] : -1;
    s_fl[i] = (headL + i < nfL) ? P.freeL[headL + i] : -1;
  }
  __syncthreads();

  // ---- candidates (extend_tracks): survivors get uv, t_total + 1, a history entry; ordered compaction ----
  auto cand_half = [&](const int (&kc)[CH], const float2 (&pc)[CH], const bool (&kp)[CH], const int (&ln)[CH], const int (&ttk)[CH]) {
    int rank[CH];
    const int n_out = pipe_rank<CH>(kp, rank, s_w);
#pragma unroll
    for (int c = 0; c < CH; c++)
      if (kp[c]) {
        const int k = kc[c];
        P.k_uv[k] = pc[c]; P.k_tt[k] = ttk[c] + 1; pipe_hist_slot(P, ln[c])[k] = pc[c]; P.k_len[k] = ln[c] + 1;
        P.cand[rank[c]] = k;
      }
    if (tid == 0) { P.cnt[C_NCAND] = n_out; P.cnt[C_NKLT] = nl + nc; }
  };
  if (which & 1) {
    if constexpr (SPLIT) {
      int kc[CH], ln[CH], ttk[CH]; float2 pc[CH]; bool kp[CH];
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const int j = tid * CH + c;
        kc[c] = 0; kp[c] = false; pc[c] = make_float2(0.f, 0.f);
        if (j < nc) { kc[c] = P.cand[j]; pc[c] = p1[nl + j]; kp[c] = pipe_inside(pc[c], W, H); }
      }
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const int j = tid * CH + c;
        ln[c] = ttk[c] = 0;
        if (j < nc) { ln[c] = P.k_len[kc[c]]; ttk[c] = P.k_tt[kc[c]]; }
      }
      cand_half(kc, pc, kp, ln, ttk);
    } else {
      cand_half(KC, qc, keepc, lenc, ttc);
    }
  }
  if (!(which & 2)) return;                          // (uniform)
  if constexpr (SPLIT) {
    lm_fields();
    __syncthreads();                                 // the row words are cleared before the survivors count into them
  }

  // ---- landmarks: survivors update their keypoint row IN PLACE (extractor.py:80-83: the deepcopy comes after) and count into their
  //      landmark row's word -- several entries may share a landmark object, which then advances by as many frames ----
#pragma unroll
  for (int c = 0; c < CH; c++)
    if (keep[c]) {
      const int k = K[c];
      P.k_uv[k] = q[c]; P.k_tt[k] = tt[c] + 1; pipe_hist_slot(P, len[c])[k] = q[c]; P.k_len[k] = len[c] + 1;
      atomicAdd(&s_row[L[c]], 1);                    // LDS (global scratch above 4 096 slots)
    }
  __syncthreads();
  int tlf[CH];                                       // t_latest of the entry's landmark object after this frame
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const int j = tid * CH + c;
    tlf[c] = (j < nl) ? tl[c] + row_ld(L[c]) : 0;
    if (keep[c]) P.l_tl[L[c]] = tlf[c];              // (the same value from every entry that shares the row)
  }
  // ---- deepcopy(k) of a survivor (extractor.py:85) matters only when the dead list holds the same keypoint object: own row, copied
  //      AFTER the in-place update (the cooperative pass below runs behind a barrier) ----
  int n_copy = 0;
  {
    bool f[CH]; int rank[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) f[c] = keep[c] && ksh[c];
    const int tot = pipe_rank<CH>(f, rank, s_w);      // (its barriers also order the reads of s_row above before the election below)
    if (headK + tot > nfK) { if (tid == 0) P.cnt[C_STATUS] |= VO_PIPE_CAPACITY; return; }
#pragma unroll
    for (int c = 0; c < CH; c++)
      if (f[c]) { s_src[rank[c]] = K[c]; K[c] = s_fk[rank[c]]; }
    n_copy = tot;
  }
  // ---- what died is deep-copied into the dead lists (pipeline.py:101-102).  One deepcopy call per list: entries that share a
  //      landmark object share its copy -> the entry with the smallest index copies the row for all of them ----
  {
    int drank[CH];
    const int n_die = pipe_rank<CH>(die, drank, s_w);
    const int room = P.N - nd0;
    bool ok[CH], lead[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) { ok[c] = die[c] && drank[c] < room; if (ok[c]) row_st(L[c], 0x7FFFFFFF); }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CH; c++) if (ok[c]) atomicMin(&s_row[L[c]], tid * CH + c);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CH; c++) lead[c] = ok[c] && row_ld(L[c]) == tid * CH + c;
    int lrank[CH], krank[CH];
    const int n_lead = pipe_rank<CH>(lead, lrank, s_w);  // (its barriers also separate the reads of s_row above from the writes below)
    const int n_ok = pipe_rank<CH>(ok, krank, s_w);
    if (headL + n_lead > nfL || headK + n_copy + n_ok > nfK) { if (tid == 0) P.cnt[C_STATUS] |= VO_PIPE_CAPACITY; return; }
#pragma unroll
    for (int c = 0; c < CH; c++)
      if (lead[c]) {
        const int nlr = s_fl[lrank[c]];
        P.l_tl[nlr] = tlf[c];
        for (int k = 0; k < 3; k++) P.l_p[3 * (size_t)nlr + k] = LATE ? P.l_p[3 * (size_t)L[c] + k] : lp[LATE ? 0 : c][k];
        row_st(L[c], nlr);
      }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CH; c++)
      if (ok[c]) {
        const int slot = n_copy + krank[c];
        s_src[slot] = K[c];
        P.dead_L[nd0 + drank[c]] = row_ld(L[c]);
        P.dead_K[nd0 + drank[c]] = s_fk[slot];
      }
    headL += n_lead; headK += n_copy + n_ok;
    n_copy += n_ok;
    if (n_die > n_ok) { overflow |= 1; inert = n_die - n_ok; }
    if (tid == 0) P.cnt[C_NDEAD] = nd0 + n_ok;
  }
  __syncthreads();                                   // the in-place row updates and the work list are complete
  // ---- trip 4: the row copies src = s_src[i] -> dst = s_fk[i], 37 values each, spread over the workgroup ----
  for (int e = tid; e < n_copy * 40; e += PIPE_TPB) {
    const int i = e / 40, f = e - i * 40;
    const int src = s_src[i], dst = s_fk[i];
    if (f < PIPE_HIST) P.k_hist[(size_t)f * P.R + dst] = P.k_hist[(size_t)f * P.R + src];
    else if (f == 32) P.k_tf[dst] = P.k_tf[src];
    else if (f == 33) P.k_tt[dst] = P.k_tt[src];
    else if (f == 34) P.k_len[dst] = P.k_len[src];
    else if (f == 35) P.k_uv[dst] = P.k_uv[src];
    else if (f == 36) P.k_first[dst] = P.k_first[src];
  }
  // ---- ordered compaction of the landmark list; the survivors are the 3D-2D correspondences of the pose stage (extractor.py:176-177) ----
  {
    int rank[CH];
    const int n_out = pipe_rank<CH>(keep, rank, s_w);
#pragma unroll
    for (int c = 0; c < CH; c++)
      if (keep[c]) {
        const int o = rank[c];
        P.lm_L[o] = L[c]; P.lm_K[o] = K[c]; P.lm_ksh[o] = 0;
        pnp_uv[2 * o] = q[c].x; pnp_uv[2 * o + 1] = q[c].y;
        for (int k = 0; k < 3; k++) pnp_X[3 * o + k] = (float)(LATE ? P.l_p[3 * (size_t)L[c] + k] : lp[LATE ? 0 : c][k]);
      }
    if (tid == 0) {
      P.cnt[C_NLM] = n_out; P.cnt[C_NPNP] = n_out;
      if (which & 1) P.cnt[C_NKLT] = nl + nc;          // (landmarks alone: the candidates' half has recorded it before it compacted its list)
      Pall.dn[DN_PNP * gridDim.x + b] = n_out;
      P.cnt[C_T] += 1; P.cnt[C_HEADK] = headK; P.cnt[C_HEADL] = headL;
      P.cnt[C_NINERT] += inert; P.cnt[C_OVERFLOW] = overflow;
    }
  }
}

// ================================================================================================
// k_pipe_prune: PnP result -> trajectory, non-inliers to the dead lists; ripe candidates -> DLT inputs
// ================================================================================================
// (same construction as k_pipe_extend: lists, then every row field and the free-list heads in one trip, the keypoint-row copies as one
//  cooperative pass.)  Dynamic LDS: int32 [N] free K rows | [N] free L rows | [N] copy sources
template <int CH>
__global__ void __launch_bounds__(PIPE_TPB) k_pipe_prune(pipe_ptrs Pall, int do_pose, int do_tri, const uint8_t* __restrict__ mask, const double* __restrict__ pnp_out,
                                                         int pnp_cap, int min_len, float* __restrict__ uv0, float* __restrict__ uv1, size_t uv_seq,
                                                         vo_dlt_cam* __restrict__ cams, int32_t* __restrict__ cam_sel) {
  extern __shared__ int32_t s_dyn[];
  __shared__ int s_w[16];
  __shared__ int s_present[PIPE_HIST];
  __shared__ int s_bad;
  const int b = blockIdx.x, tid = threadIdx.x;
  const pipe_ptrs P = pipe_select(Pall, b);
  if (P.cnt[C_STATUS]) return;
  int32_t* const s_fk = s_dyn;
  int32_t* const s_fl = s_fk + P.N;
  int32_t* const s_src = s_fl + P.N;
  mask += (size_t)b * pnp_cap; pnp_out += 8 * (size_t)b;
  uv0 += (size_t)b * uv_seq; uv1 += (size_t)b * uv_seq; cam_sel += (size_t)b * P.N;
  // ---- trip 1: counters, the pose ----
  const int nl = P.cnt[C_NLM], nc = P.cnt[C_NCAND], nd0 = P.cnt[C_NDEAD], t = P.cnt[C_T];
  int headK = P.cnt[C_HEADK], headL = P.cnt[C_HEADL];
  const int nfK = P.cnt[C_NFREEK], nfL = P.cnt[C_NFREEL];
  int overflow = P.cnt[C_OVERFLOW];
  const double n_inl = do_pose ? pnp_out[7] : 4.0;
  if (tid == 0) s_bad = 0;
  if (tid < PIPE_HIST) s_present[tid] = 0;
  __syncthreads();
  if (do_pose && !(n_inl >= 4.0)) { if (tid == 0) P.cnt[C_STATUS] |= VO_PIPE_LOST; return; }      // cv2.solvePnPRansac found nothing: the reference crashes here
  // ---- trip 2: lists, inlier mask ----
  int L[CH], K[CH], KC[CH];
  bool in[CH], out[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const int j = tid * CH + c;
    L[c] = K[c] = KC[c] = 0; in[c] = out[c] = false;
    if (do_pose && j < nl) { L[c] = P.lm_L[j]; K[c] = P.lm_K[j]; in[c] = mask[j] != 0; out[c] = !in[c]; }
    if (do_tri && j < nc) KC[c] = P.cand[j];
  }
  // ---- trip 3: row fields, free-list heads ----
  constexpr bool LATE = CH >= 8;                       // (as in k_pipe_extend: 224 bytes of scratch per lane at CH = 8 with the positions prefetched)
  constexpr int CE = LATE ? 1 : CH;
  int tlo[CH], ttc[CH], tfc[CH];
  double lp[CE][3];
  float2 fst[CH], cur[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const int j = tid * CH + c;
    tlo[c] = ttc[c] = tfc[c] = 0; fst[c] = cur[c] = make_float2(0.f, 0.f);
    if (!LATE) lp[c][0] = lp[c][1] = lp[c][2] = 0.0;
    if (out[c]) { tlo[c] = P.l_tl[L[c]]; if (!LATE) for (int k = 0; k < 3; k++) lp[c][k] = P.l_p[3 * (size_t)L[c] + k]; }
    if (do_tri && j < nc) { ttc[c] = P.k_tt[KC[c]]; tfc[c] = P.k_tf[KC[c]]; fst[c] = P.k_first[KC[c]]; cur[c] = P.k_uv[KC[c]]; }
  }
  if (do_pose)
    for (int i = tid; i < P.N; i += PIPE_TPB) {
      s_fk[i] = (headK + i < nfK) ? P.freeK[headK + i] : -1;
      s_fl[i] = (headL + i < nfL) ? P.freeL[headL + i] : -1;
    }
  __syncthreads();
  int n_copy = 0;
  if (do_pose) {
    // non-inliers: deepcopy per entry (pipeline.py:133-134), every one gets its own L and K copy
    int drank[CH], rank[CH];
    const int n_out = pipe_rank<CH>(out, drank, s_w);
    const int room = P.N - nd0;
    const int n_ok = n_out < room ? n_out : room;
    if (headL + n_ok > nfL || headK + n_ok > nfK) { if (tid == 0) P.cnt[C_STATUS] |= VO_PIPE_CAPACITY; return; }
#pragma unroll
    for (int c = 0; c < CH; c++)
      if (out[c] && drank[c] < room) {
        const int nlr = s_fl[drank[c]], nk = s_fk[drank[c]];
        P.l_tl[nlr] = tlo[c];
        for (int k = 0; k < 3; k++) P.l_p[3 * (size_t)nlr + k] = LATE ? P.l_p[3 * (size_t)L[c] + k] : lp[LATE ? 0 : c][k];
        s_src[drank[c]] = K[c];
        P.dead_L[nd0 + drank[c]] = nlr; P.dead_K[nd0 + drank[c]] = nk;
      }
    n_copy = n_ok;
    headL += n_ok; headK += n_ok;
    if (n_out > n_ok) overflow |= 1;
    const int n_in = pipe_rank<CH>(in, rank, s_w);     // (its barriers also publish the work list)
#pragma unroll
    for (int c = 0; c < CH; c++) if (in[c]) { P.lm_L[rank[c]] = L[c]; P.lm_K[rank[c]] = K[c]; P.lm_ksh[rank[c]] = 0; }
    if (tid == 0) {
      P.cnt[C_NLM] = n_in; P.cnt[C_NDEAD] = nd0 + n_ok; P.cnt[C_NINERT] += n_out - n_ok;
      // trajectory.append(t, [R(rvec) | tvec]) (extractor.py:186-191, pipeline.py:140)
      double R[9];
      pipe_rodrigues(pnp_out, R);
      double* Hd = P.H + 12 * (size_t)(t & (PIPE_HIST - 1));
      for (int r = 0; r < 3; r++) { for (int k = 0; k < 3; k++) Hd[4 * r + k] = R[3 * r + k]; Hd[4 * r + 3] = pnp_out[3 + r]; }
    }
    // the keypoint-row copies src = s_src[i] -> dst = s_fk[i]
    for (int e = tid; e < n_copy * 40; e += PIPE_TPB) {
      const int i = e / 40, f = e - i * 40;
      const int src = s_src[i], dst = s_fk[i];
      if (f < PIPE_HIST) P.k_hist[(size_t)f * P.R + dst] = P.k_hist[(size_t)f * P.R + src];
      else if (f == 32) P.k_tf[dst] = P.k_tf[src];
      else if (f == 33) P.k_tt[dst] = P.k_tt[src];
      else if (f == 34) P.k_len[dst] = P.k_len[src];
      else if (f == 35) P.k_uv[dst] = P.k_uv[src];
      else if (f == 36) P.k_first[dst] = P.k_first[src];
    }
  }
  __syncthreads();                                     // the new pose is in the ring
  if (do_tri) {
    // triangulate_tracks (extractor.py:202-203): candidates that reached min_track_length leave the list whether or not they succeed.
    // They are triangulated in groups by birth frame, each against the pose of its birth frame (:210-220); a group is named by its AGE
    // a = t - t_first (the slot distance in the 32-frame trajectory ring)
    int age[CH]; bool ripe[CH], wait[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const int j = tid * CH + c;
      age[c] = 0; ripe[c] = wait[c] = false;
      if (j < nc) { ripe[c] = ttc[c] >= min_len; wait[c] = !ripe[c]; age[c] = t - tfc[c]; }
    }
    int rr[CH], wr[CH];
    const int n_ripe = pipe_rank<CH>(ripe, rr, s_w);
    const int n_wait = pipe_rank<CH>(wait, wr, s_w);
#pragma unroll
    for (int c = 0; c < CH; c++) {
      if (ripe[c]) {
        if (age[c] < 0 || age[c] >= PIPE_HIST || age[c] > t) s_bad = 1;       // the birth pose has left the trajectory ring
        else s_present[age[c]] = 1;
        uv0[2 * rr[c]] = fst[c].x; uv0[2 * rr[c] + 1] = fst[c].y; uv1[2 * rr[c]] = cur[c].x; uv1[2 * rr[c] + 1] = cur[c].y;
        P.ripe[rr[c]] = KC[c]; cam_sel[rr[c]] = age[c] & (PIPE_HIST - 1);
      }
      if (wait[c]) P.cand[wr[c]] = KC[c];
    }
    __syncthreads();
    if (tid < PIPE_HIST && s_present[tid] && !s_bad) {
      // P = K @ H[:3] rounded to float32 for the DLT (extractor.py:268-269), float64 for the filter statistics
      const double* H0 = P.H + 12 * (size_t)((t - tid) & (PIPE_HIST - 1)); const double* H1 = P.H + 12 * (size_t)(t & (PIPE_HIST - 1));
      vo_dlt_cam& a = cams[(size_t)b * PIPE_HIST + tid];
      for (int r = 0; r < 3; r++)
        for (int col = 0; col < 4; col++) {
          double s0 = 0, s1 = 0;
          for (int k = 0; k < 3; k++) { s0 += P.Kc[3 * r + k] * H0[4 * k + col]; s1 += P.Kc[3 * r + k] * H1[4 * k + col]; }
          a.M0[4 * r + col] = s0; a.M1[4 * r + col] = s1; a.P0[4 * r + col] = (float)s0; a.P1[4 * r + col] = (float)s1;
        }
      for (int k = 0; k < 4; k++) a.H1z[k] = H1[8 + k];
    }
    if (tid == 0) {
      P.cnt[C_NCAND] = n_wait; P.cnt[C_NRIPE] = n_ripe;
      Pall.dn[DN_RIPE * gridDim.x + b] = n_ripe;
      if (n_ripe > 0 && s_bad) { P.cnt[C_STATUS] |= VO_PIPE_GROUPS; P.cnt[C_NRIPE] = 0; Pall.dn[DN_RIPE * gridDim.x + b] = 0; }
    }
  }
  if (tid == 0) { P.cnt[C_HEADK] = headK; P.cnt[C_HEADL] = headL; P.cnt[C_OVERFLOW] = overflow; }
}

// The order in which the reference walks the birth-frame groups of the ripe candidates: `for t_first in set([k.t_first for k in ...])`
// (extractor.py:210-211) iterates a CPython set of small non-negative ints, i.e. the slots of its open-addressing table (Objects/setobject.c):
// 8 slots, x4 whenever fill * 5 >= mask * 3 (32 at the 5th key, 128 at the 19th; a resize re-inserts in slot order), slot = key & mask
// (hash(i) = i), on a collision up to 9 linear probes while they stay inside the table, then i = (5 i + 1 + perturb) & mask with
// perturb >>= 5.  {2, 9, 16, 8} is walked 16, 9, 2, 8.  Keys are inserted in order of first appearance in the ripe list (s_any[age] = that
// index, 0x7FFFFFFF = no such group); oracle/pipe_oracle.py:cpython_set_order is the same walk, checked against the interpreter.
// -> s_order[0 .. *s_n) = the ages in walk order.  Called by the whole workgroup (barriers inside); one group (every frame of a steady run)
// takes the short way.
__device__ inline void pipe_set_insert(int* s_tab, int mask, int key) {
  int perturb = key, i = key & mask;
  for (;;) {
    if (s_tab[i] < 0) { s_tab[i] = key; return; }
    if (i + 9 <= mask)
      for (int j = i + 1; j <= i + 9; j++) if (s_tab[j] < 0) { s_tab[j] = key; return; }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}

__device__ __forceinline__ void pipe_set_order(const int* s_any, int* s_order, int* s_tab, int* s_tmp, int* s_n, int t) {
  const int tid = threadIdx.x;
  if (tid < 128) s_tab[tid] = -1;
  int rank = -1;
  if (tid < PIPE_HIST && s_any[tid] != 0x7FFFFFFF) {     // rank of this age by first appearance = its insertion index
    rank = 0;
    for (int a = 0; a < PIPE_HIST; a++) rank += (s_any[a] < s_any[tid]) ? 1 : 0;
  }
  __syncthreads();
  if (rank >= 0) s_order[rank] = tid;
  __syncthreads();
  if (tid == 0) {
    int n = 0;
    for (int a = 0; a < PIPE_HIST; a++) n += (s_any[a] != 0x7FFFFFFF) ? 1 : 0;
    if (n > 1) {
      int mask = 7, fill = 0;
      for (int k = 0; k < n; k++) {
        pipe_set_insert(s_tab, mask, t - s_order[k]);    // key = birth frame
        fill++;
        if (fill * 5 >= mask * 3) {                      // grow: the keys leave in slot order and are inserted again
          int m = 0;
          for (int i = 0; i <= mask; i++) if (s_tab[i] >= 0) { s_tmp[m++] = s_tab[i]; s_tab[i] = -1; }
          int size = 8;
          while (size <= fill * 4) size <<= 1;
          mask = size - 1;
          for (int i = 0; i < m; i++) pipe_set_insert(s_tab, mask, s_tmp[i]);
        }
      }
      int m = 0;
      for (int i = 0; i <= mask; i++) if (s_tab[i] >= 0) s_tmp[m++] = t - s_tab[i];
      for (int i = 0; i < m; i++) s_order[i] = s_tmp[i];
    }
    *s_n = n;
  }
  __syncthreads();
}

// One entry j of the final lists -> the bundle-adjustment problem (bundle_adjuster.py:153-176: point, observations from the keypoint history;
// unused slots: zeros / NaN) and the resident point set [landmark keypoints | candidates] of the next frame's tracker.  -> observations written
__device__ __forceinline__ int pipe_emit_entry(const pipe_ptrs& P, int j, int nl, int nc, int t, int do_adjust, int Wn, int Nba, double* __restrict__ x0,
                                               double* __restrict__ obs, float2* __restrict__ out, int Kr, int len, int tl, const double (&p3d)[3],
                                               float2 uvk, float2 cuvj) {
  int nobs = 0;
  if (do_adjust && j < Nba) {
    double* const pts0 = x0 + 6 * (size_t)Wn;
    if (j < nl) {
      for (int k = 0; k < 3; k++) pts0[3 * (size_t)j + k] = p3d[k];
      for (int s = 0; s < Wn; s++) {
        const int idx = (t - s) - tl + len - 1;                      // k.uv_history[(t_now - s) - l.t_latest + len - 1] (:56-59, :156)
        double2 v = make_double2(__builtin_nan(""), __builtin_nan(""));
        if (idx >= 0 && idx <= len - 1 && idx >= len - PIPE_HIST) { const float2 h = pipe_hist_slot(P, idx)[Kr]; v = make_double2((double)h.x, (double)h.y); nobs++; }
        reinterpret_cast<double2*>(obs)[(size_t)s * Nba + j] = v;
      }
    } else {
      for (int k = 0; k < 3; k++) pts0[3 * (size_t)j + k] = 0.0;
      for (int s = 0; s < Wn; s++) reinterpret_cast<double2*>(obs)[(size_t)s * Nba + j] = make_double2(__builtin_nan(""), __builtin_nan(""));
    }
  }
  if (j < nl) out[j] = uvk;
  if (j < nc) out[nl + j] = cuvj;
  return nobs;
}

// k_pipe_problem: the last phase of k_pipe_promote for tables above 2 048 slots, one lane per list entry over N / 256 workgroups per sequence
// (behind k_pipe_promote on the same stream: the lists and counters are final)
__global__ void __launch_bounds__(256) k_pipe_problem(pipe_ptrs Pall, int do_adjust, int Wn, double* __restrict__ x0, double* __restrict__ obs, size_t x_stride,
                                                      size_t obs_stride, int Nba, float* __restrict__ pts, size_t pts_seq) {
  const int b = blockIdx.y;
  const pipe_ptrs P = pipe_select(Pall, b);
  if (P.cnt[C_STATUS]) return;
  const int nl = P.cnt[C_NLM], nc = P.cnt[C_NCAND], t = P.cnt[C_T];
  const int j = blockIdx.x * 256 + threadIdx.x;
  x0 += (size_t)b * x_stride; obs += (size_t)b * obs_stride;
  int Kr = 0, len = 0, tl = 0;
  double p3d[3] = {0.0, 0.0, 0.0};
  float2 uvk = make_float2(0.f, 0.f), cv = make_float2(0.f, 0.f);
  if (j < nl) {
    const int Lr = P.lm_L[j];
    Kr = P.lm_K[j];
    len = P.k_len[Kr]; tl = P.l_tl[Lr]; uvk = P.k_uv[Kr];
    for (int k = 0; k < 3; k++) p3d[k] = P.l_p[3 * (size_t)Lr + k];
  }
  if (j < nc) cv = P.k_uv[P.cand[j]];
  int nobs = (j < P.N) ? pipe_emit_entry(P, j, nl, nc, t, do_adjust, Wn, Nba, x0, obs, reinterpret_cast<float2*>(vo_seq(pts, pts_seq, b)), Kr, len, tl, p3d, uvk, cv) : 0;
  for (int o = 32; o > 0; o >>= 1) nobs += __shfl_xor(nobs, o);
  if (do_adjust && nobs && (threadIdx.x & 63) == 0) atomicAdd(&P.cnt[C_NOBS], nobs);
}

// ================================================================================================
// k_pipe_promote: triangulation filters + gate + promotion; then the selection half of BundleAdjuster.adjust
// ================================================================================================
// (Constructed like k_pipe_extend: trip 1 counters; trip 2 every list and per-candidate input, the free-row window and the whole trajectory
//  ring into LDS; trip 3 the row fields of the dead entries, of the EXISTING state entries -- thread j owns final entry j, and entries are
//  only appended -- and of the candidates; trips 4 / 5 the rows of appended entries and the history entries of the observation table.
//  The landmark list is kept in LDS while it grows, "this landmark row is in the state's list" is a byte per row in LDS.)
// Dynamic LDS: int32 [N] landmark rows of the list | [N] keypoint rows of the list | [N] free L rows | bytes [R] row-in-list marks
template <int CH>
__global__ void __launch_bounds__(PIPE_TPB) k_pipe_promote(pipe_ptrs Pall, int do_tri, int do_adjust, const float* __restrict__ X4, const double* __restrict__ depth1,
                                                           const double* __restrict__ reproj, size_t slab_seq, int x4_stride, double max_err,
                                                           double min_angle, const int32_t* __restrict__ cam_sel, int Wn, int resurrect, double* __restrict__ x0, double* __restrict__ obs,
                                                           size_t x_stride, size_t obs_stride, int Nba, float* __restrict__ pts, size_t pts_seq) {
  extern __shared__ int32_t s_dyn[];
  __shared__ int s_w[16];
  __shared__ int s_first[PIPE_HIST], s_gate[PIPE_HIST];
  __shared__ int s_any[PIPE_HIST], s_order[PIPE_HIST], s_tab[128], s_tmp[PIPE_HIST], s_norder;   // the walk order of the birth groups (pipe_set_order)
  __shared__ double s_p3[PIPE_HIST][3];
  __shared__ double s_H[PIPE_HIST * 12];
  __shared__ int s_nobs;
  const int b = blockIdx.x, tid = threadIdx.x;
  const pipe_ptrs P = pipe_select(Pall, b);
  if (P.cnt[C_STATUS]) return;
  int32_t* const s_lmL = s_dyn;
  int32_t* const s_lmK = s_lmL + P.N;
  int32_t* const s_fl = s_lmK + P.N;
  uint8_t* const s_mark = reinterpret_cast<uint8_t*>(s_fl + P.N);
  X4 = vo_seq(X4, slab_seq, b); depth1 = vo_seq(depth1, slab_seq, b); reproj = vo_seq(reproj, slab_seq, b);
  cam_sel += (size_t)b * P.N;
  x0 += (size_t)b * x_stride; obs += (size_t)b * obs_stride;
  // ---- trip 1: counters ----
  const int t = P.cnt[C_T], nc = P.cnt[C_NCAND], n_ripe = do_tri ? P.cnt[C_NRIPE] : 0, nd0 = P.cnt[C_NDEAD];
  const int nl0 = P.cnt[C_NLM];
  int nl = nl0;
  const int headL0 = P.cnt[C_HEADL];
  const int nfL = P.cnt[C_NFREEL];
  int overflow = P.cnt[C_OVERFLOW];
  if (tid == 0) s_nobs = 0;
  if (tid < PIPE_HIST) { s_first[tid] = 0x7FFFFFFF; s_gate[tid] = 0; s_any[tid] = 0x7FFFFFFF; }
  __syncthreads();
  // ---- trip 2: lists, candidates of the triangulation, free rows, trajectory ring ----
  // LATE (CH >= 4, tables above 2 048 slots): what a thread holds per entry -- CH of everything -- no longer fits the 128 registers a lane
  // of a 1 024-lane workgroup gets (CH = 8: 708 bytes of scratch per lane, the kernel 177 us in a batch of 32).  There the fields that
  // are only needed at the end (position, pixel, history length, t_latest of the existing entries; the triangulated point of a ripe
  // candidate) are fetched where they are used, one dependent trip later, in blocks of four entries.
  constexpr bool LATE = CH >= 4;
  constexpr int CE = LATE ? 1 : CH;                   // entries whose prefetched fields are kept (LATE: none; the arrays shrink to one dummy)
  bool kept[CH]; float pt[CE][3]; int age[CH], rk[CH];
  int DL[CH], DK[CH], eL[CE], eK[CE], cK[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const int j = tid * CH + c;
    kept[c] = false; age[c] = 0; rk[c] = 0;
    if (!LATE) { pt[c][0] = pt[c][1] = pt[c][2] = 0.f; eL[c] = eK[c] = 0; }
    DL[c] = DK[c] = cK[c] = 0;
    if (j < n_ripe) {
      if (!LATE) {
        const float w4 = X4[(size_t)3 * x4_stride + j];
        for (int k = 0; k < 3; k++) pt[c][k] = X4[(size_t)k * x4_stride + j] / w4;        // numpy float32 divide (extractor.py:271)
      }
      kept[c] = depth1[j] > 0.0 && reproj[j] < max_err;                                    // triangulate.py:87-111
      age[c] = cam_sel[j]; rk[c] = P.ripe[j];
    }
    if (do_adjust && j < nd0) { DL[c] = P.dead_L[j]; DK[c] = P.dead_K[j]; }
    if (j < nl0) {
      const int l = P.lm_L[j], k = P.lm_K[j];
      s_lmL[j] = l; s_lmK[j] = k;
      if (!LATE) { eL[c] = l; eK[c] = k; }
    }
    if (j < nc) cK[c] = P.cand[j];
  }
  for (int i = tid; i < P.N; i += PIPE_TPB) s_fl[i] = (headL0 + i < nfL) ? P.freeL[headL0 + i] : -1;
  for (int i = tid; i < PIPE_HIST * 12; i += PIPE_TPB) s_H[i] = P.H[i];
  // ---- trip 3: row fields ----
  int dtl[CH], dlen[CH], elen[CE], etl[CE];
  double ep[CE][3];
  float2 euv[CE], cuv[CE];
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const int j = tid * CH + c;
    dtl[c] = dlen[c] = 0;
    if (do_adjust && j < nd0) { dtl[c] = P.l_tl[DL[c]]; dlen[c] = P.k_len[DK[c]]; }
    if (!LATE) {
      elen[c] = etl[c] = 0; ep[c][0] = ep[c][1] = ep[c][2] = 0.0; euv[c] = cuv[c] = make_float2(0.f, 0.f);
      if (j < nl0) {
        elen[c] = P.k_len[eK[c]]; etl[c] = P.l_tl[eL[c]]; euv[c] = P.k_uv[eK[c]];
        for (int k = 0; k < 3; k++) ep[c][k] = P.l_p[3 * (size_t)eL[c] + k];
      }
      if (j < nc) cuv[c] = P.k_uv[cK[c]];
    }
  }
  __syncthreads();
  // which dead entries lie inside the window (bundle_adjuster.py:132-150): decided here, so that t_latest / history length of the dead
  // entries are not carried through the triangulation
  bool win[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const int j = tid * CH + c;
    win[c] = do_adjust && resurrect && j < nd0 && (t - (dtl[c] - (dlen[c] - 1))) < Wn;
  }
  int n_new = 0, fl_off = 0;
  if (do_tri && n_ripe > 0) {
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const int j = tid * CH + c;
      if (kept[c]) atomicMin(&s_first[age[c]], j);
      if (j < n_ripe) atomicMin(&s_any[age[c]], j);      // first appearance of the birth frame in the ripe list, kept or not
    }
    __syncthreads();
    pipe_set_order(s_any, s_order, s_tab, s_tmp, &s_norder, t);
#pragma unroll
    for (int c = 0; c < CH; c++)
      if (kept[c] && s_first[age[c]] == tid * CH + c) {
        if (LATE) {
          const int j = tid * CH + c;
          const float w4 = X4[(size_t)3 * x4_stride + j];
          for (int k = 0; k < 3; k++) s_p3[age[c]][k] = (double)(X4[(size_t)k * x4_stride + j] / w4);
        } else {
          for (int k = 0; k < 3; k++) s_p3[age[c]][k] = (double)pt[c][k];
        }
      }
    __syncthreads();
    if (tid < PIPE_HIST && s_first[tid] != 0x7FFFFFFF) {
      // the reference's "bearing angle" of the group's FIRST landmark (extractor.py:231-240): a = Frobenius norm of the 4x4 relative
      // transform, b = |H0 [p; 0]|, c = |H1 [p; 0]|, law of cosines in degrees; NaN rejects
      const double p3[3] = {s_p3[tid][0], s_p3[tid][1], s_p3[tid][2]};
      const double* H0 = s_H + 12 * (size_t)((t - tid) & (PIPE_HIST - 1)); const double* H1 = s_H + 12 * (size_t)(t & (PIPE_HIST - 1));
      double inv0[12];                                   // inv([R | t]) = [R^T | -R^T t]
      for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) inv0[4 * r + k] = H0[4 * k + r];
        inv0[4 * r + 3] = -(H0[r] * H0[3] + H0[4 + r] * H0[7] + H0[8 + r] * H0[11]);
      }
      double a2 = 1.0;                                   // bottom row [0 0 0 1]
      for (int r = 0; r < 3; r++)
        for (int col = 0; col < 4; col++) {
          double sacc = 0;
          for (int k = 0; k < 3; k++) sacc += H1[4 * r + k] * inv0[4 * k + col];
          if (col == 3) sacc += H1[4 * r + 3];
          a2 += sacc * sacc;
        }
      double b2 = 0, c2 = 0;
      for (int r = 0; r < 3; r++) {
        const double v0 = H0[4 * r] * p3[0] + H0[4 * r + 1] * p3[1] + H0[4 * r + 2] * p3[2];
        const double v1 = H1[4 * r] * p3[0] + H1[4 * r + 1] * p3[1] + H1[4 * r + 2] * p3[2];
        b2 += v0 * v0; c2 += v1 * v1;
      }
      const double a = sqrt(a2), bb = sqrt(b2), cc = sqrt(c2);
      const double theta = acos((bb * bb + cc * cc - a * a) / (2.0 * bb * cc)) * 57.29577951308232;
      s_gate[tid] = (theta > min_angle) ? 1 : 0;         // false for NaN
    }
    __syncthreads();
    // accepted groups are appended in the order the reference walks them -- `for t_first in set(...)` (extractor.py:210-211): CPython's
    // iteration order of a set of small ints, not ascending (pipe_set_order) --, each in list order (extractor.py:238-240)
    int room = P.N - nl - nc;
    if (room < 0) room = 0;
    const int n_order = s_norder;
    for (int g = 0; g < n_order; g++) {
      const int a = s_order[g];
      if (!s_gate[a]) continue;                          // uniform: s_gate and s_order are shared
      bool f[CH]; int rank[CH];
#pragma unroll
      for (int c = 0; c < CH; c++) f[c] = kept[c] && age[c] == a;
      const int n_grp = pipe_rank<CH>(f, rank, s_w);
      int n_take = n_grp < room ? n_grp : room;
      if (n_grp > n_take) overflow |= 2;
      if (headL0 + fl_off + n_take > nfL) { if (tid == 0) P.cnt[C_STATUS] |= VO_PIPE_CAPACITY; return; }
#pragma unroll
      for (int c = 0; c < CH; c++)
        if (f[c] && rank[c] < n_take) {
          const int nlr = s_fl[fl_off + rank[c]], o = nl + rank[c];
          P.l_tl[nlr] = t;                                                                   // Landmark(t_curr, p, des) (extractor.py:274-275)
          float p3f[3];
          if (LATE) {
            const int j = tid * CH + c;
            const float w4 = X4[(size_t)3 * x4_stride + j];
            for (int k = 0; k < 3; k++) p3f[k] = X4[(size_t)k * x4_stride + j] / w4;       // numpy float32 divide (extractor.py:271)
          } else {
            for (int k = 0; k < 3; k++) p3f[k] = pt[c][k];
          }
          for (int k = 0; k < 3; k++) P.l_p[3 * (size_t)nlr + k] = (double)p3f[k];
          s_lmL[o] = nlr; s_lmK[o] = rk[c];
          P.lm_L[o] = nlr; P.lm_K[o] = rk[c]; P.lm_ksh[o] = 0;
        }
      fl_off += n_take; nl += n_take; n_new += n_take; room -= n_take;
    }
  }
  __syncthreads();
  int n_res = 0, nd = nd0, n_inert = 0;
  if (do_adjust) {
    // ---- dead landmarks whose track lies inside the window are appended to the state's lists as the same objects (bundle_adjuster.py:132-150) ----
    bool take[CH], stay[CH];
    int wr[CH];
    const int n_win = pipe_rank<CH>(win, wr, s_w);
    int room = P.N - nl - nc;
    if (room < 0) room = 0;
    n_res = n_win < room ? n_win : room;
    if (n_win > n_res) overflow |= 4;
#pragma unroll
    for (int c = 0; c < CH; c++) {
      take[c] = win[c] && wr[c] < n_res;
      if (take[c]) {
        const int o = nl + wr[c];
        s_lmL[o] = DL[c]; s_lmK[o] = DK[c];
        P.lm_L[o] = DL[c]; P.lm_K[o] = DK[c]; P.lm_ksh[o] = 1;
      }
      const int j = tid * CH + c;
      if (j < nd0 && !take[c]) s_mark[DL[c]] = 0;
    }
    nl += n_res;
    __syncthreads();
    // an entry that stays dead is kept while it may still be resurrected: the window test holds, or its L row is in the state's list
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const int j = tid * CH + c;
      if (j < nl) s_mark[s_lmL[j]] = 1;
      // (capacity policy only: an entry inside the window that found no room stays a candidate for resurrection, and so does every entry
      //  that shares its landmark row -- once it returns, the row's t_latest advances again for them too)
      if (j < nd0 && win[c] && !take[c]) s_mark[DL[c]] = 1;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CH; c++) { const int j = tid * CH + c; stay[c] = j < nd0 && !take[c] && (win[c] || s_mark[DL[c]] == 1); }
    int tr[CH], sr[CH];
    const int n_take = pipe_rank<CH>(take, tr, s_w);
    const int n_stay = pipe_rank<CH>(stay, sr, s_w);
#pragma unroll
    for (int c = 0; c < CH; c++) {
      if (take[c]) { P.dead_L[tr[c]] = DL[c]; P.dead_K[tr[c]] = DK[c]; }
      if (stay[c]) { P.dead_L[n_take + sr[c]] = DL[c]; P.dead_K[n_take + sr[c]] = DK[c]; }
    }
    nd = n_take + n_stay; n_inert = nd0 - nd;
  }
  __syncthreads();                                     // the list in LDS and the new landmark rows are complete
  // ---- trip 4: rows of the entries appended above (the existing ones were fetched in trip 3), the bundle-adjustment problem and the
  //      resident point set of the final lists (pipe_emit_entry; folded in here: one launch less on the frame's critical chain).
  //      LATE: that bulk work -- 8 entries per lane through ONE compute unit's memory pipes, ~60 us of this kernel at 8 192 slots --
  //      is left to k_pipe_problem, a launch of N / 256 workgroups per sequence right behind this one ----
  int nobs = 0;
  if constexpr (!LATE) {
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const int j = tid * CH + c;
      int Kr = eK[c], len = elen[c], tl = etl[c];
      double p3d[3] = {ep[c][0], ep[c][1], ep[c][2]};
      float2 uvk = euv[c];
      if (j >= nl0 && j < nl) {
        const int Lr = s_lmL[j];
        Kr = s_lmK[j];
        len = P.k_len[Kr]; tl = P.l_tl[Lr]; uvk = P.k_uv[Kr];
        for (int k = 0; k < 3; k++) p3d[k] = P.l_p[3 * (size_t)Lr + k];
      }
      nobs += pipe_emit_entry(P, j, nl, nc, t, do_adjust, Wn, Nba, x0, obs, reinterpret_cast<float2*>(vo_seq(pts, pts_seq, b)), Kr, len, tl, p3d, uvk, cuv[c]);
    }
  }
  if (do_adjust) {
    if (nobs) atomicAdd(&s_nobs, nobs);
    if (tid < Wn) {
      double* po = x0 + 6 * tid;
      const int tt = t - tid;
      if (tt >= 0 && tid < PIPE_HIST) {                              // poses missing at the start of a sequence stay zero (:169-171)
        const double* Hs = s_H + 12 * (size_t)(tt & (PIPE_HIST - 1));
        const double R[9] = {Hs[0], Hs[1], Hs[2], Hs[4], Hs[5], Hs[6], Hs[8], Hs[9], Hs[10]};
        pipe_log_so3(R, po);
        po[3] = Hs[3]; po[4] = Hs[7]; po[5] = Hs[11];
      } else {
        for (int k = 0; k < 6; k++) po[k] = 0.0;
      }
    }
  }
  __syncthreads();                                     // s_nobs
  if (tid == 0) {
    P.cnt[C_NLM] = nl; P.cnt[C_NNEW] = n_new; P.cnt[C_HEADL] = headL0 + fl_off; P.cnt[C_OVERFLOW] = overflow;
    if (do_adjust) { P.cnt[C_NDEAD] = nd; P.cnt[C_NINERT] += n_inert; P.cnt[C_NRES] = n_res; P.cnt[C_NOBS] = s_nobs; }
    P.cnt[C_NPTS] = nl + nc; Pall.dn[DN_PTS * gridDim.x + b] = nl + nc; Pall.dn[DN_ROOM * gridDim.x + b] = max(P.N - nl - nc, 0) + 1;
  }
}

// ================================================================================================
// k_pipe_writeback: solution -> landmark rows and trajectory (bundle_adjuster.py:197-213)
// ================================================================================================

template <int CH>
__global__ void __launch_bounds__(PIPE_TPB) k_pipe_writeback(pipe_ptrs Pall, int do_adjust, const uint8_t* __restrict__ pub, size_t pub_bytes, const double* __restrict__ x0,
                                                             size_t x_stride, int Wn, vo_pipe_record* __restrict__ rec) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const pipe_ptrs P = pipe_select(Pall, b);
  // (C_STATUS, C_NLM, C_T, C_NOBS and the landmark list are final before the fork: k_pipe_spawn, which may run beside this kernel on the
  //  side stream, touches the candidate list, fresh keypoint rows and the free lists)
  const int nl = P.cnt[C_NLM], t = P.cnt[C_T];
  if (do_adjust && !P.cnt[C_STATUS]) {
    const ba_state* st = reinterpret_cast<const ba_state*>(pub + (size_t)b * pub_bytes);      // (vo_internal.h: shared with vo_ba.hip)
    // nothing observed: the adapter skips the solve and writes x0 back (bundle_adjuster.py would hand scipy an empty problem)
    const double* x = (P.cnt[C_NOBS] > 0) ? reinterpret_cast<const double*>(pub + (size_t)b * pub_bytes + VO_BA_PUB_HEADER) : x0 + (size_t)b * x_stride;
    const double* xp = x + 6 * (size_t)Wn;
    // entries that share a landmark row: the reference assigns in list order, the LAST one wins (:197-201)
#pragma unroll
    for (int c = 0; c < CH; c++) { const int j = tid * CH + c; if (j < nl) st_i32(&P.scr[P.lm_L[j]], -1); }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CH; c++) { const int j = tid * CH + c; if (j < nl) atomicMax(&P.scr[P.lm_L[j]], j); }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const int j = tid * CH + c;
      if (j < nl) { const int Lr = P.lm_L[j]; if (ld_i32(&P.scr[Lr]) == j) for (int k = 0; k < 3; k++) P.l_p[3 * (size_t)Lr + k] = xp[3 * (size_t)j + k]; }
    }
    if (tid < Wn && t - tid >= 0 && tid < PIPE_HIST) {
      double R[9];
      pipe_rodrigues(x + 6 * tid, R);
      double* Hd = P.H + 12 * (size_t)((t - tid) & (PIPE_HIST - 1));
      for (int r = 0; r < 3; r++) { for (int k = 0; k < 3; k++) Hd[4 * r + k] = R[3 * r + k]; Hd[4 * r + 3] = x[6 * tid + 3 + r]; }
    }
    if (tid == 0) {
      vo_pipe_record& r = rec[b];
      const bool solved = P.cnt[C_NOBS] > 0;
      r.ba_landmarks = nl; r.ba_observations = P.cnt[C_NOBS];
      r.ba_iters = solved ? st->iter : 0; r.ba_accepted = solved ? st->accepted : 0; r.ba_status = solved ? st->status : 0;
      r.ba_done = solved ? st->done : 1; r.ba_cost0 = solved ? st->cost0 : 0.0; r.ba_cost = solved ? st->cost : 0.0;
    }
  }
  __syncthreads();                                   // the window's poses are in the ring
  if (tid == 0) {
    // the frame's pose as the adjustment leaves it, and the oldest pose of the window: the next adjust no longer touches that one
    vo_pipe_record& r = rec[b];
    const double* Hs = P.H + 12 * (size_t)(t & (PIPE_HIST - 1));
    for (int k = 0; k < 12; k++) r.H[k] = Hs[k];
    const int tf = t - (Wn - 1);
    r.t_final = tf >= 0 ? tf : -1;
    const double* Hf = P.H + 12 * (size_t)((tf >= 0 ? tf : t) & (PIPE_HIST - 1));
    for (int k = 0; k < 12; k++) r.H_final[k] = Hf[k];
  }
}

// ================================================================================================
// k_pipe_spawn: corners -> candidates; free rows; the frame's record
// ================================================================================================
template <int CH>
__global__ void __launch_bounds__(PIPE_TPB) k_pipe_spawn(pipe_ptrs Pall, int do_detect, const uint32_t* __restrict__ st_scalars, const float* __restrict__ st_out,
                                                         float* __restrict__ pts, size_t slab_seq, int max_new, const double* __restrict__ pnp_out,
                                                         const int32_t* __restrict__ pnp_ctrl, size_t pnp_ctrl_stride, vo_pipe_record* __restrict__ rec, int rebuild) {
  __shared__ int s_w[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const pipe_ptrs P = pipe_select(Pall, b);
  vo_pipe_record& r = rec[b];
  const int status = P.cnt[C_STATUS];
  const int t = P.cnt[C_T], nl = P.cnt[C_NLM];
  int nc = P.cnt[C_NCAND];
  int overflow = P.cnt[C_OVERFLOW], n_det = 0;
  __syncthreads();
  if (!status && do_detect) {
    const uint32_t* sc = vo_seq(st_scalars, slab_seq, b);
    const float2* corners = reinterpret_cast<const float2*>(vo_seq(st_out, slab_seq, b));
    float2* out = reinterpret_cast<float2*>(vo_seq(pts, slab_seq, b));
    const uint32_t ndet = sc[2];
    int m = (ndet == 0xFFFFFFFFu) ? 0 : (int)ndet;
    if (ndet == 0xFFFFFFFFu) overflow |= 16;
    if (m > max_new) m = max_new;
    int room = P.N - nl - nc;
    if (room < 0) room = 0;
    if (m > room) { m = room; overflow |= 8; }
    const int headK = P.cnt[C_HEADK], nfK = P.cnt[C_NFREEK];
    if (headK + m > nfK) { if (tid == 0) P.cnt[C_STATUS] |= VO_PIPE_CAPACITY; m = 0; }
    for (int i = tid; i < m; i += PIPE_TPB) {
      // Keypoint(t_first = t, t_total = 1, uv_first = uv = corner, uv_history = [corner]) (extractor.py:127-130)
      const int k = P.freeK[headK + i];
      const float2 q = corners[i];
      P.k_tf[k] = t; P.k_tt[k] = 1; P.k_len[k] = 1; P.k_uv[k] = q; P.k_first[k] = q; pipe_hist_slot(P, 0)[k] = q;
      P.cand[nc + i] = k; out[nl + nc + i] = q;
    }
    n_det = m; nc += m;
    __syncthreads();
    if (tid == 0) { P.cnt[C_NCAND] = nc; P.cnt[C_NPTS] = nl + nc; Pall.dn[DN_PTS * gridDim.x + b] = nl + nc; P.cnt[C_HEADK] = headK + m; }
  }
  __syncthreads();
  // ---- rows no list refers to go back to the free lists (ascending).  Both mark arrays live in LDS (one byte per row), a thread owns
  //      RPT consecutive rows, so the two free lists cost two block scans and no global round trip besides reading the lists (the earlier
  //      form marked in global scratch and scanned the R rows 1 024 at a time: 16 scans, 8 global phases) ----
  const int nd = P.cnt[C_NDEAD];
  if (rebuild) {       // (uniform.  Stage-wise callers may keep the free lists until the frame's last stage: VO_PIPE_KEEP_FREE_LISTS)
    extern __shared__ uint32_t s_mark_dyn[];                      // [K | L][R / 4] words: one byte per row (2 x 32 KB at 8 192 slots)
    const int mw = (P.R + 3) / 4;
    uint8_t* const mk = reinterpret_cast<uint8_t*>(s_mark_dyn);
    uint8_t* const ml = reinterpret_cast<uint8_t*>(s_mark_dyn + mw);
    for (int i = tid; i < 2 * mw; i += PIPE_TPB) s_mark_dyn[i] = 0;
    __syncthreads();
    for (int j = tid; j < P.N; j += PIPE_TPB) {
      if (j < nc) mk[P.cand[j]] = 1;
      if (j < nl) { mk[P.lm_K[j]] = 1; ml[P.lm_L[j]] = 1; }
      if (j < nd) { mk[P.dead_K[j]] = 1; ml[P.dead_L[j]] = 1; }
    }
    __syncthreads();
    const int rpt = (P.R + PIPE_TPB - 1) / PIPE_TPB;             // rows per thread (<= 32)
    const int r0 = min(tid * rpt, P.R), r1 = min(r0 + rpt, P.R);
    for (int pass = 0; pass < 2; pass++) {
      const uint8_t* m = pass == 0 ? mk : ml;
      int32_t* fl = pass == 0 ? P.freeK : P.freeL;
      int cnt = 0;
      for (int i = r0; i < r1; i++) cnt += m[i] == 0 ? 1 : 0;
      int total;
      int pos = pipe_scan_count(cnt, s_w, total);
      for (int i = r0; i < r1; i++) if (m[i] == 0) fl[pos++] = i;
      if (tid == 0) { P.cnt[pass == 0 ? C_NFREEK : C_NFREEL] = total; P.cnt[pass == 0 ? C_HEADK : C_HEADL] = 0; }
    }
  }
  if (tid == 0) {
    r.t = t; r.status = P.cnt[C_STATUS]; r.overflow = overflow;
    r.n_landmarks = nl; r.n_candidates = nc; r.n_dead = nd; r.n_dead_total = nd + P.cnt[C_NINERT];
    r.n_tracked = P.cnt[C_NKLT];
    const double n_in = pnp_out[8 * (size_t)b + 7];
    r.pnp_inliers = (n_in == n_in) ? (int)n_in : 0; r.pnp_hypotheses = pnp_ctrl[(size_t)b * pnp_ctrl_stride + 1];
    r.pnp_bound_reached = pnp_ctrl[(size_t)b * pnp_ctrl_stride + 2];
    r.n_ripe = P.cnt[C_NRIPE]; r.n_new = P.cnt[C_NNEW]; r.n_resurrected = P.cnt[C_NRES]; r.n_detected = n_det;
    P.cnt[C_NDET] = n_det; P.cnt[C_OVERFLOW] = overflow;       // (H, H_final, t_final and the ba_* fields: k_pipe_writeback)
  }
}

// resident point set = every keypoint of the state (landmark entries first): the exclusion discs of this frame's re-detection (extractor.py:103-107
// with pipeline.py:160: state._landmarks_kp + state._candidates_kp AFTER adjust has appended the resurrected entries), then -- with the corners
// k_pipe_spawn appends -- the next frame's KLT input.  Also run after the tables were written from the host (reset = 1).
__global__ void __launch_bounds__(PIPE_TPB) k_pipe_dense(pipe_ptrs Pall, float* __restrict__ pts, size_t slab_seq, int reset) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const pipe_ptrs P = pipe_select(Pall, b);
  if (!reset && P.cnt[C_STATUS]) return;
  const int nl = P.cnt[C_NLM], nc = P.cnt[C_NCAND];
  float2* out = reinterpret_cast<float2*>(vo_seq(pts, slab_seq, b));
  for (int j = tid; j < P.N; j += PIPE_TPB) {
    if (j < nl) out[j] = P.k_uv[P.lm_K[j]];
    if (j < nc) out[nl + j] = P.k_uv[P.cand[j]];
  }
  if (tid == 0) { P.cnt[C_NPTS] = nl + nc; Pall.dn[DN_PTS * gridDim.x + b] = nl + nc; Pall.dn[DN_ROOM * gridDim.x + b] = max(P.N - nl - nc, 0) + 1; if (reset) { Pall.dn[DN_PNP * gridDim.x + b] = 0; Pall.dn[DN_RIPE * gridDim.x + b] = 0; } }
}

// ================================================================================================
// host
// ================================================================================================
bool vo_pipe_busy(const vo_ctx* c) { return c->pipe && c->pipe->enq != c->pipe->fetched; }

void vo_pipe_destroy(vo_ctx* c) {
  if (!c->pipe) return;
  vo_pipe_ws* w = c->pipe;
  void* bufs[] = {w->tab_slab, w->d_gather, w->d_gather_rows, w->d_ripe, w->d_freeK, w->d_freeL, w->d_scr, w->d_dn, w->d_cam_sel, w->d_cams, w->d_K, w->d_rec};
  for (void* p : bufs) if (p) (void)hipFree(p);
  if (w->h_rec) (void)hipHostFree(w->h_rec);
  for (hipEvent_t e : w->ev) if (e) (void)hipEventDestroy(e);
  if (w->ev_track) (void)hipEventDestroy(w->ev_track);
  delete w;
  c->pipe = nullptr;
  vo_ba_set_live(c, nullptr, 0);       // (the counters the bundle adjustment looked at are gone)
}

extern "C" int32_t vo_pipe_default_params(vo_pipe_params* p) {
  if (!p) return VO_E_INVALID;
  memset(p, 0, sizeof(*p));
  p->ba_window = 4; p->min_track_length = 3; p->mask_radius = 7; p->max_new = 1000; p->pnp_blind_batches = 4;
  p->max_reproj_err = 2.0; p->min_bearing_angle = 0.5; p->resurrect = 1;
  vo_klt_default_params(&p->klt); vo_st_default_params(&p->st); vo_ba_default_params(&p->ba); vo_pnp_default_params(&p->pnp);
  p->st.min_distance = 7.0;
  p->ba_budget = p->ba.max_iters;
  return VO_OK;
}

static int32_t pipe_create(vo_ctx* c, const double* K, const vo_pipe_params* prm, bool* replaced);

extern "C" int32_t vo_pipe_create(vo_ctx* c, const double* K, const vo_pipe_params* prm) {
  if (!c) return VO_E_INVALID;
  bool replaced = false;                          // rejected arguments leave an existing pipeline alone
  const int32_t r = pipe_create(c, K, prm, &replaced);
  if (r != VO_OK && replaced) vo_pipe_destroy(c); // never leave a half-built workspace behind (vo_last_error keeps the reason)
  return r;
}

static int32_t pipe_create(vo_ctx* c, const double* K, const vo_pipe_params* prm, bool* replaced) {
  VO_CHECK(c, K, VO_E_INVALID, "null K");
  vo_pipe_params def;
  if (!prm) { vo_pipe_default_params(&def); prm = &def; }
  VO_CHECK(c, prm->ba_window >= 1 && prm->ba_window <= 20, VO_E_INVALID, "ba_window must be 1..20");
  VO_CHECK(c, c->max_pts <= PIPE_TPB * PIPE_CH, VO_E_CAPACITY, "the pipeline tables hold at most 8192 keypoints per sequence (max_pts)");
  VO_CHECK(c, prm->min_track_length >= 1 && prm->max_new >= 0 && prm->pnp_blind_batches >= 1 && prm->pnp_blind_batches <= 64, VO_E_INVALID, "bad parameters");
  VO_CHECK(c, prm->ba.max_iters >= 0 && prm->ba.max_iters <= 1000 && prm->ba_budget >= 0 && prm->ba_budget <= prm->ba.max_iters, VO_E_INVALID,
           "ba_budget must be 0..ba.max_iters");
  VO_HIP(c, hipSetDevice(c->device));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  vo_pipe_destroy(c);
  *replaced = true;
  vo_pipe_ws* w = new vo_pipe_ws();
  c->pipe = w;
  w->prm = *prm;
  w->N = c->max_pts; w->R = 4 * c->max_pts;
  const size_t B = c->batch, N = w->N, R = w->R;
  const size_t sz[VO_PIPE_N_TABLES] = {4 * R, 4 * R, 4 * R, 8 * R, 8 * R, 8 * R * PIPE_HIST, 4 * R, 24 * R, 4 * N, 4 * N, 4 * N, 4 * N, 4 * N, 4 * N,
                                       4 * PIPE_NCNT, 8 * 12 * PIPE_HIST};
  {
    size_t off[VO_PIPE_N_TABLES + 1] = {};
    for (int i = 0; i < VO_PIPE_N_TABLES; i++) { w->tab_bytes[i] = sz[i]; off[i + 1] = off[i] + ((sz[i] * B + 255) & ~(size_t)255); }
    VO_HIP(c, hipMalloc((void**)&w->tab_slab, off[VO_PIPE_N_TABLES]));
    VO_HIP(c, hipMemsetAsync(w->tab_slab, 0, off[VO_PIPE_N_TABLES], c->stream));
    for (int i = 0; i < VO_PIPE_N_TABLES; i++) w->tab[i] = w->tab_slab + off[i];
    w->lists_bytes = off[VO_PIPE_N_TABLES] - off[VO_PIPE_CAND];
  }
  VO_HIP(c, hipMalloc((void**)&w->d_gather, 288 * N * B));
  VO_HIP(c, hipMalloc((void**)&w->d_gather_rows, 4 * N * B));
  VO_HIP(c, hipMalloc((void**)&w->d_ripe, 4 * N * B));
  VO_HIP(c, hipMalloc((void**)&w->d_freeK, 4 * R * B));
  VO_HIP(c, hipMalloc((void**)&w->d_freeL, 4 * R * B));
  VO_HIP(c, hipMalloc((void**)&w->d_scr, 4 * R * B));
  VO_HIP(c, hipMalloc((void**)&w->d_cam_sel, 4 * N * B));
  VO_HIP(c, hipMemsetAsync(w->d_cam_sel, 0, 4 * N * B, c->stream));
  VO_HIP(c, hipMalloc((void**)&w->d_cams, sizeof(vo_dlt_cam) * PIPE_HIST * B));
  VO_HIP(c, hipMemsetAsync(w->d_cams, 0, sizeof(vo_dlt_cam) * PIPE_HIST * B, c->stream));
  VO_HIP(c, hipMalloc((void**)&w->d_dn, 4 * 4 * B));
  VO_HIP(c, hipMemsetAsync(w->d_dn, 0, 4 * 4 * B, c->stream));
  VO_HIP(c, hipMalloc((void**)&w->d_K, 72 * B));
  VO_HIP(c, hipMalloc((void**)&w->d_rec, sizeof(vo_pipe_record) * B));
  VO_HIP(c, hipMemsetAsync(w->d_rec, 0, sizeof(vo_pipe_record) * B, c->stream));
  VO_HIP(c, hipHostMalloc((void**)&w->h_rec, sizeof(vo_pipe_record) * B * VO_PIPE_INFLIGHT, hipHostMallocDefault));
  const unsigned fl = hipEventDisableTiming | (vo_blocking_sync() ? hipEventBlockingSync : 0u);
  for (int i = 0; i < VO_PIPE_INFLIGHT; i++) VO_HIP(c, hipEventCreateWithFlags(&w->ev[i], fl));
  VO_HIP(c, hipEventCreateWithFlags(&w->ev_track, hipEventDisableTiming));
  // k_pipe_extend keeps 28 bytes of LDS per table slot (112 KB at 4 096 slots: above the default limit of a launch)
  VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_extend<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 28 * PIPE_TPB * 4));
  VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_extend<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 28 * PIPE_TPB * 2));
  VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_promote<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * PIPE_TPB * 4));
  // 8 192 slots: extend 96 KB (the three windows; the row words are global), prune 96 KB, promote 128 KB, spawn 64 KB of the CU's 160 KB
  VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_extend<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 12 * PIPE_TPB * 8));
  VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_prune<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 12 * PIPE_TPB * 8));
  VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_promote<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * PIPE_TPB * 8));
  VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_spawn<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PIPE_TPB * 8));
  VO_HIP(c, hipMemcpyAsync(w->d_K, K, 72 * B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  int32_t r = vo_ba_reserve(c, K, prm->ba_window, w->N);
  if (r != VO_OK) return r;
  // (the adjustment walks the landmark slots in use, not the table: pipe_step names the counters before every enqueue -- an uploaded
  //  problem, vo_ba_upload*, forgets them)
  r = vo_pnp_reserve(c, K);
  if (r != VO_OK) return r;
  r = vo_st_prepare(c, &w->prm.st);
  if (r != VO_OK) return r;
  c->n_resident = c->max_pts;
  // (the per-sequence counters -- live points DN_PTS, free slots + 1 DN_ROOM -- are handed to the KLT / re-detection launches by pipe_step
  //  itself: the context's other resident entry points never see them)
  return vo_pipe_commit(c);
}

extern "C" int32_t vo_pipe_table_bytes(vo_ctx* c, int32_t which, uint64_t* bytes) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->pipe && which >= 0 && which < VO_PIPE_N_TABLES && bytes, VO_E_INVALID, "vo_pipe_create first / bad table");
  *bytes = (uint64_t)c->pipe->tab_bytes[which] * (uint64_t)c->batch;
  return VO_OK;
}

extern "C" int32_t vo_pipe_table_write(vo_ctx* c, int32_t which, const void* src) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->pipe && which >= 0 && which < VO_PIPE_N_TABLES && src, VO_E_INVALID, "vo_pipe_create first / bad table");
  VO_CHECK(c, c->pipe->enq == c->pipe->fetched, VO_E_STATE, "fetch the steps in flight first");
  VO_HIP(c, hipSetDevice(c->device));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  VO_HIP(c, hipMemcpy(c->pipe->tab[which], src, c->pipe->tab_bytes[which] * c->batch, hipMemcpyHostToDevice));
  return VO_OK;
}

extern "C" int32_t vo_pipe_table_read(vo_ctx* c, int32_t which, void* dst) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->pipe && which >= 0 && which < VO_PIPE_N_TABLES && dst, VO_E_INVALID, "vo_pipe_create first / bad table");
  VO_HIP(c, hipSetDevice(c->device));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  VO_HIP(c, hipMemcpy(dst, c->pipe->tab[which], c->pipe->tab_bytes[which] * c->batch, hipMemcpyDeviceToHost));
  return VO_OK;
}

// The object boundary (vo_mi355x/lazy.py: the reference's Extractor / BundleAdjuster interface over these tables) reads, after a stage,
// the LISTS -- tables VO_PIPE_CAND .. VO_PIPE_POSES as they lie in the slab, each [batch][...] and padded to 256 bytes: `bytes` from
// vo_pipe_lists_bytes, offsets = the running sum of the padded table sizes -- in ONE copy, and single object rows by index.
extern "C" int32_t vo_pipe_lists_bytes(vo_ctx* c, uint64_t* bytes) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->pipe && bytes, VO_E_INVALID, "vo_pipe_create first / null output");
  *bytes = c->pipe->lists_bytes;
  return VO_OK;
}

extern "C" int32_t vo_pipe_lists_read(vo_ctx* c, void* dst) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->pipe && dst, VO_E_INVALID, "vo_pipe_create first / null output");
  VO_CHECK(c, c->pipe->enq == c->pipe->fetched, VO_E_STATE, "fetch the steps in flight first");
  VO_HIP(c, hipSetDevice(c->device));
  VO_HIP(c, hipMemcpyAsync(dst, c->pipe->tab[VO_PIPE_CAND], c->pipe->lists_bytes, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}

// packed object rows: K row = {i32 t_first, t_total, hist_len, 0; f32 uv[2], uv_first[2]; f32 hist[32][2]} (288 bytes, ring slot order),
// L row = {i32 t_latest, 0; f64 p[3]} (32 bytes)
__global__ void k_pipe_gather(pipe_ptrs Pall, int kind, const int32_t* __restrict__ rows, int n, int cap, uint8_t* __restrict__ out) {
  const int b = blockIdx.y;
  const pipe_ptrs P = pipe_select(Pall, b);
  rows += (size_t)b * cap;
  if (kind == 0) {
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5), f = threadIdx.x & 31;     // 32 lanes per row
    if (i >= n) return;
    const int r = rows[i];
    uint8_t* o = out + ((size_t)b * cap + i) * 288;
    if (r < 0 || r >= P.R) { if (f == 0) reinterpret_cast<int32_t*>(o)[0] = 0x7FFFFFFF; return; }
    reinterpret_cast<float2*>(o + 32)[f] = P.k_hist[(size_t)f * P.R + r];
    if (f == 0) { int32_t* h = reinterpret_cast<int32_t*>(o); h[0] = P.k_tf[r]; h[1] = P.k_tt[r]; h[2] = P.k_len[r]; h[3] = 0; }
    if (f == 1) reinterpret_cast<float2*>(o + 16)[0] = P.k_uv[r];
    if (f == 2) reinterpret_cast<float2*>(o + 24)[0] = P.k_first[r];
  } else {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = rows[i];
    uint8_t* o = out + ((size_t)b * cap + i) * 32;
    if (r < 0 || r >= P.R) { reinterpret_cast<int32_t*>(o)[0] = 0x7FFFFFFF; return; }
    reinterpret_cast<int32_t*>(o)[0] = P.l_tl[r]; reinterpret_cast<int32_t*>(o)[1] = 0;
    for (int k = 0; k < 3; k++) reinterpret_cast<double*>(o + 8)[k] = P.l_p[3 * (size_t)r + k];
  }
}

// rows [batch][n] (row indices of one kind: 0 = K rows, 1 = L rows), n <= max_pts -> out [batch][n][288 | 32 bytes]; synchronous
extern "C" int32_t vo_pipe_rows_read(vo_ctx* c, int32_t kind, const int32_t* rows, int32_t n, void* out) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->pipe && rows && out && (kind == 0 || kind == 1), VO_E_INVALID, "vo_pipe_create first / bad arguments");
  vo_pipe_ws* w = c->pipe;
  VO_CHECK(c, n >= 0 && n <= w->N, VO_E_CAPACITY, "at most max_pts rows per call");
  VO_CHECK(c, w->enq == w->fetched, VO_E_STATE, "fetch the steps in flight first");
  if (n == 0) return VO_OK;
  VO_HIP(c, hipSetDevice(c->device));
  const size_t rec = kind == 0 ? 288 : 32;
  VO_HIP(c, hipMemcpy2DAsync(w->d_gather_rows, 4 * (size_t)w->N, rows, 4 * (size_t)n, 4 * (size_t)n, c->batch, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_pipe_gather, dim3(kind == 0 ? vo_div_up(n, 8) : vo_div_up(n, 256), c->batch), dim3(256), 0, c->stream, pipe_make(w), kind,
                     w->d_gather_rows, n, w->N, w->d_gather);
  VO_HIP(c, hipGetLastError());
  VO_HIP(c, hipMemcpy2DAsync(out, rec * n, w->d_gather, rec * w->N, rec * n, c->batch, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}

// the consensus set of the last POSE stage: mask [batch][n] over the landmark list as it was BEFORE the pruning (n = the record's pnp
// input count, <= max_pts); synchronous
extern "C" int32_t vo_pipe_inliers_read(vo_ctx* c, uint8_t* mask, int32_t n) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->pipe && mask, VO_E_INVALID, "vo_pipe_create first / null output");
  vo_pipe_ws* w = c->pipe;
  VO_CHECK(c, n >= 0 && n <= w->N, VO_E_CAPACITY, "at most max_pts entries");
  VO_CHECK(c, w->enq == w->fetched, VO_E_STATE, "fetch the steps in flight first");
  if (n == 0) return VO_OK;
  VO_HIP(c, hipSetDevice(c->device));
  vo_pnp_view pv;
  { const int32_t r = vo_pnp_get_view(c, &pv); if (r != VO_OK) return r; }
  VO_HIP(c, hipMemcpy2DAsync(mask, n, pv.mask, pv.cap, n, c->batch, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}

// entries per thread of the list kernels: the lists hold at most max_pts entries
#define PIPE_DISPATCH_LDS(KERNEL, LDS, ...)                                                                                          \
  do {                                                                                                                               \
    if (w->N <= PIPE_TPB) hipLaunchKernelGGL(KERNEL<1>, dim3(c->batch), dim3(PIPE_TPB), LDS, c->stream, __VA_ARGS__);                 \
    else if (w->N <= 2 * PIPE_TPB) hipLaunchKernelGGL(KERNEL<2>, dim3(c->batch), dim3(PIPE_TPB), LDS, c->stream, __VA_ARGS__);        \
    else if (w->N <= 4 * PIPE_TPB) hipLaunchKernelGGL(KERNEL<4>, dim3(c->batch), dim3(PIPE_TPB), LDS, c->stream, __VA_ARGS__);        \
    else hipLaunchKernelGGL(KERNEL<8>, dim3(c->batch), dim3(PIPE_TPB), LDS, c->stream, __VA_ARGS__);                                  \
  } while (0)
#define PIPE_DISPATCH(KERNEL, ...) PIPE_DISPATCH_LDS(KERNEL, 0, __VA_ARGS__)

static void pipe_launch_spawn(vo_ctx* c, int do_detect, int rebuild = 1) {
  vo_pipe_ws* w = c->pipe;
  vo_pnp_view pv;
  (void)vo_pnp_get_view(c, &pv);
  PIPE_DISPATCH_LDS(k_pipe_spawn, 2 * sizeof(uint32_t) * (size_t)((w->R + 3) / 4), pipe_make(w), do_detect, vo_slab<const uint32_t>(c, c->off_st_scalars),
                     vo_slab<const float>(c, c->off_st_out), vo_slab<float>(c, vo_off_p(c)), c->slab_seq, w->prm.max_new, pv.out, pv.ctrl,
                pv.ctrl_stride, w->d_rec, rebuild);
}

extern "C" int32_t vo_pipe_commit(vo_ctx* c) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->pipe, VO_E_STATE, "vo_pipe_create first");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  vo_pipe_ws* w = c->pipe;
  hipLaunchKernelGGL(k_pipe_dense, dim3(c->batch), dim3(PIPE_TPB), 0, c->stream, pipe_make(w), vo_slab<float>(c, vo_off_p(c)), c->slab_seq, 1);
  pipe_launch_spawn(c, 0);                      // free lists (and a record of the seeded state)
  { vo_ba_view bv; const int32_t rb = vo_ba_get_view(c, &bv); if (rb != VO_OK) return rb;
    const pipe_ptrs P = pipe_make(w);
    PIPE_DISPATCH(k_pipe_writeback, P, 0, bv.pub, bv.pub_bytes, bv.x0, bv.x_stride, bv.W, w->d_rec); }
  VO_HIP(c, hipGetLastError());
  VO_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}

extern "C" int32_t vo_pipe_set_ba_budget(vo_ctx* c, int32_t budget) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->pipe, VO_E_STATE, "vo_pipe_create first");
  VO_CHECK(c, budget >= 0 && budget <= c->pipe->prm.ba.max_iters, VO_E_INVALID, "budget must be 0..ba.max_iters");
  c->pipe->prm.ba_budget = budget;
  return VO_OK;
}

static int32_t pipe_step(vo_ctx* c, int32_t frame_idx, int32_t stages, bool main_dirty);

#define PIPE_FRAME_FROM_HOST (-2)      // internal frame index: the step's images are in c->d_host_raw[c->pipe_host_slot] behind ev_h2d

static int32_t pipe_step_entry(vo_ctx* c, int32_t frame_idx, int32_t stages);

extern "C" int32_t vo_pipe_step(vo_ctx* c, int32_t frame_idx, int32_t stages) {
  if (!c) return VO_E_INVALID;
  return pipe_step_entry(c, frame_idx < 0 ? -1 : frame_idx, stages);
}

// the same step with this frame's images handed over by the host (Pipeline.step(img), pipeline.py:98,171-172): the upload runs on the copy stream
// (k_gather_frames for page-locked images), the pyramid + tracking of the step on the side stream behind it, exactly like a resident frame
extern "C" int32_t vo_pipe_step_host(vo_ctx* c, const uint8_t* const* frames, int32_t stride, int32_t stages) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->pipe, VO_E_STATE, "vo_pipe_create first");
  VO_CHECK(c, frames != nullptr && stride >= c->width, VO_E_INVALID, "bad frame pointers / stride");
  VO_CHECK(c, (stages & VO_PIPE_TRACK) != 0, VO_E_INVALID, "a step that takes a frame tracks it (VO_PIPE_TRACK)");
  for (int b = 0; b < c->batch; b++) VO_CHECK(c, frames[b] != nullptr, VO_E_INVALID, "null frame pointer");
  VO_CHECK(c, c->pipe->enq - c->pipe->fetched < VO_PIPE_INFLIGHT, VO_E_STATE, "vo_pipe_fetch the oldest step first");
  VO_HIP(c, hipSetDevice(c->device));
  c->pipe_host_slot ^= 1;
  { const int32_t ru = vo_host_frames_upload(c, frames, stride, c->pipe_host_slot); if (ru != VO_OK) return ru; }
  return pipe_step_entry(c, PIPE_FRAME_FROM_HOST, stages);
}

static int32_t pipe_step_entry(vo_ctx* c, int32_t frame_idx, int32_t stages) {
  VO_CHECK(c, c->pipe, VO_E_STATE, "vo_pipe_create first");
  VO_CHECK(c, c->pipe->enq - c->pipe->fetched < VO_PIPE_INFLIGHT, VO_E_STATE, "vo_pipe_fetch the oldest step first");
  VO_HIP(c, hipSetDevice(c->device));
  // the closed loop's chain (PnP, the adjustment) lives on the ctx stream: it gets every compute unit (the pipelined frame step of a batch
  // confines that stream to 224 of them for its tracker launches, vo_set_side_stream)
  // -- the layout stays what vo_set_side_stream made it: it is SUSPENDED here and the next vo_frame_step_* puts it back (vo_step_layout reports
  // what is in effect at the moment).  vo_set_side_stream / vo_set_tuning refuse while pipe steps are in flight, so a masked stream is never met
  // with steps of this loop still on it
  if (c->stream_reserve > 0 || c->ba_wide_groups > 0) {
    VO_CHECK(c, c->pipe->enq == c->pipe->fetched && c->steps_enq == c->steps_fetched, VO_E_STATE, "steps in flight on the gated stream layout");
    const int32_t rr = vo_main_stream_reserve(c, 0);
    if (rr != VO_OK) return rr;
    c->ba_wide_groups = 0; c->ba_wide_recorded = false; c->layout_suspended = true;
  }
  const bool dirty = c->main_dirty;               // something other than a pipe step has used the ctx stream since the last one
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  c->main_dirty = false;
  hipStream_t const main_stream = c->stream;
  c->in_step = true;                              // the stage calls below must not wait for the side streams on the host
  const int32_t r = pipe_step(c, frame_idx, stages, dirty);
  c->in_step = false;
  c->stream = main_stream;                        // (an early return may have left the side stream selected)
  return r;
}

static int32_t pipe_step(vo_ctx* c, int32_t frame_idx, int32_t stages, bool main_dirty) {
  vo_pipe_ws* w = c->pipe;
  const vo_pipe_params& prm = w->prm;
  const pipe_ptrs P = pipe_make(w);
  const int B = c->batch;
  int32_t r;
  vo_pnp_view pv; vo_ba_view bv;
  r = vo_pnp_get_view(c, &pv); if (r != VO_OK) return r;
  r = vo_ba_get_view(c, &bv); if (r != VO_OK) return r;
  // Stream layout with a side stream (vo_set_side_stream != 0).  The loop-carried chain of the closed loop is
  //   extend -> PnP -> prune -> DLT -> promote -> bundle adjustment -> write-back -> extend of the next frame (it copies landmark positions);
  // everything else hangs off it and runs on the side stream, in that stream's order:
  //   [after promote]  re-detection -> spawn (candidate list, free rows, the list half of the record)
  //   [next step]      pyramid + KLT of the next frame (they need the spawned candidates, not the adjustment)
  // so the adjustment of frame t runs beside the re-detection of frame t AND the tracking of frame t + 1 as soon as two steps are in
  // flight.  The main stream waits for the side stream twice per step (before extend, before the record leaves), so after every call
  // it is downstream of all side work.  A frame the caller pushed itself (frame_idx < 0) is tracked on the main stream as before.
  const bool side = c->side_stream != 0 && c->stream2 != nullptr;
  hipStream_t const main_stream = c->stream;
  const int halves = (stages & (VO_PIPE_TRACK_CANDIDATES | VO_PIPE_TRACK_LANDMARKS)) ? (((stages & VO_PIPE_TRACK_CANDIDATES) ? 1 : 0) | ((stages & VO_PIPE_TRACK_LANDMARKS) ? 2 : 0)) : 3;
  // the landmarks' half on its own reads the positions a TRACK | TRACK_CANDIDATES call left in the point buffer: without that call before it
  // k_pipe_extend would take stale positions for tracked ones
  if (!(stages & VO_PIPE_TRACK) && (stages & VO_PIPE_TRACK_LANDMARKS))
    VO_CHECK(c, halves == 2 && w->lm_half_pending, VO_E_STATE, "VO_PIPE_TRACK_LANDMARKS alone needs a VO_PIPE_TRACK | VO_PIPE_TRACK_CANDIDATES call before it");
  if (!(stages & VO_PIPE_TRACK) && (stages & VO_PIPE_TRACK_CANDIDATES))
    return vo_fail(c, VO_E_STATE, "pipe_step: VO_PIPE_TRACK_CANDIDATES without VO_PIPE_TRACK (the candidates' half follows the tracking in the same call)");
  if (stages & (VO_PIPE_TRACK | VO_PIPE_TRACK_LANDMARKS)) w->lm_half_pending = (stages & VO_PIPE_TRACK) && halves == 1;
  if (stages & VO_PIPE_TRACK) {
    const bool from_host = frame_idx == PIPE_FRAME_FROM_HOST, has_frame = frame_idx >= 0 || from_host;
    if (frame_idx >= 0) VO_CHECK(c, c->d_seq && frame_idx < c->seq_n, VO_E_STATE, "no resident sequence / bad frame index");
    VO_CHECK(c, c->n_pushed + (has_frame ? 1 : 0) >= 2, VO_E_STATE, "tracking needs two frames in the frame store");
    const bool track_side = side && has_frame;
    r = VO_OK;
    if (track_side && main_dirty) {
      // between two pipe steps the side stream needs no order against the main stream beyond the events below (that is the overlap); after
      // anything else (vo_frame_push_resident, a table write, another stage call: all asynchronous on the ctx stream) it starts behind it
      VO_HIP(c, hipEventRecord(c->ev_fork, c->stream));
      VO_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    }
    // (the pyramid on a stream of its own, ahead of the side stream's re-detection: measured, slower -- every cross-stream event costs more
    //  than the five launches it would overlap; one context of 96 sequences 37.5 k against 39.0 k frames/s, one sequence 2 320 against 2 715)
    if (track_side) c->stream = c->stream2;
    if (from_host) {
      const int slot = c->pipe_host_slot;
      const hipError_t ew = hipStreamWaitEvent(c->stream, c->ev_h2d[slot], 0);
      if (ew != hipSuccess) { c->stream = main_stream; VO_HIP(c, ew); }
      r = vo_build_pyramid(c, c->d_host_raw[slot], (size_t)c->width * c->height, nullptr);
      if (r == VO_OK && hipEventRecord(c->ev_raw_free[slot], c->stream) == hipSuccess) c->raw_free_recorded[slot] = true;
    } else if (frame_idx >= 0) {
      const size_t fr = (size_t)c->width * c->height;
      r = vo_build_pyramid(c, c->d_seq + (size_t)frame_idx * fr, fr * c->seq_n, nullptr);
    }
    if (r == VO_OK) r = vo_klt_track_resident_counts(c, w->N, &prm.klt, w->d_dn + DN_PTS * B);
    c->stream = main_stream;
    if (track_side) {                                  // joined on every path
      const hipError_t e1 = hipEventRecord(w->ev_track, c->stream2);
      const hipError_t e2 = hipStreamWaitEvent(c->stream, w->ev_track, 0);
      if (r == VO_OK) { VO_HIP(c, e1); VO_HIP(c, e2); }
    }
    if (r != VO_OK) return r;
  }
  // the keep rule / bookkeeping on the tracked point set: with TRACK, or alone (VO_PIPE_TRACK_LANDMARKS after a TRACK | VO_PIPE_TRACK_CANDIDATES call)
  if ((stages & VO_PIPE_TRACK) || (stages & (VO_PIPE_TRACK_CANDIDATES | VO_PIPE_TRACK_LANDMARKS))) {
    const size_t lds = sizeof(int32_t) * ((w->N > 4 * PIPE_TPB ? 0 : (size_t)w->R) + 3 * (size_t)w->N);     // (above 4 096 slots the row words are global)
    auto launch = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(c->batch), dim3(PIPE_TPB), lds, c->stream, P, vo_slab<const float>(c, vo_off_p(c)), c->slab_seq,
                         c->width, c->height, pv.X, pv.uv, pv.cap, halves, const_cast<uint8_t*>(pv.mask));
    };
    if (w->N <= PIPE_TPB) launch(k_pipe_extend<1>);
    else if (w->N <= 2 * PIPE_TPB) launch(k_pipe_extend<2>);
    else if (w->N <= 4 * PIPE_TPB) launch(k_pipe_extend<4>);
    else launch(k_pipe_extend<8>);
  }
  if (stages & VO_PIPE_POSE) {
    r = vo_pnp_enqueue_counts(c, &prm.pnp, prm.pnp_blind_batches, w->d_dn + DN_PNP * B);
    if (r != VO_OK) return r;
  }
  if (stages & (VO_PIPE_POSE | VO_PIPE_TRIANGULATE))
  {
    const size_t lds = sizeof(int32_t) * 3 * (size_t)w->N;
    auto launch = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(c->batch), dim3(PIPE_TPB), lds, c->stream, P, (stages & VO_PIPE_POSE) ? 1 : 0, (stages & VO_PIPE_TRIANGULATE) ? 1 : 0,
                         pv.mask, pv.out, pv.cap, prm.min_track_length, c->d_uv0, c->d_uv1, (size_t)c->max_pts * 2, w->d_cams, w->d_cam_sel);
    };
    if (w->N <= PIPE_TPB) launch(k_pipe_prune<1>);
    else if (w->N <= 2 * PIPE_TPB) launch(k_pipe_prune<2>);
    else if (w->N <= 4 * PIPE_TPB) launch(k_pipe_prune<4>);
    else launch(k_pipe_prune<8>);
  }
  if (stages & VO_PIPE_TRIANGULATE) {
    r = vo_dlt_enqueue_counts(c, w->N, w->d_dn + DN_RIPE * B, w->d_cams, w->d_cam_sel, PIPE_HIST);
    if (r != VO_OK) return r;
  }
  if (stages & (VO_PIPE_TRIANGULATE | VO_PIPE_ADJUST))
  {
    const size_t lds = sizeof(int32_t) * 3 * (size_t)w->N + (size_t)w->R;
    auto launch = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(c->batch), dim3(PIPE_TPB), lds, c->stream, P, (stages & VO_PIPE_TRIANGULATE) ? 1 : 0, (stages & VO_PIPE_ADJUST) ? 1 : 0,
                         vo_slab<const float>(c, c->off_X4), vo_slab<const double>(c, c->off_depth), vo_slab<const double>(c, c->off_reproj), c->slab_seq,
                         w->N, prm.max_reproj_err, prm.min_bearing_angle, w->d_cam_sel, bv.W, prm.resurrect, bv.x0, bv.obs, bv.x_stride, bv.obs_stride, bv.N,
                         vo_slab<float>(c, vo_off_p(c)), c->slab_seq);
    };
    if (w->N <= PIPE_TPB) launch(k_pipe_promote<1>);
    else if (w->N <= 2 * PIPE_TPB) launch(k_pipe_promote<2>);
    else if (w->N <= 4 * PIPE_TPB) launch(k_pipe_promote<4>);
    else launch(k_pipe_promote<8>);
    if (w->N > 2 * PIPE_TPB)     // (the CH >= 4 kernels leave the problem / point-set phase to a wide launch)
      hipLaunchKernelGGL(k_pipe_problem, dim3(vo_div_up(w->N, 256), c->batch), dim3(256), 0, c->stream, P, (stages & VO_PIPE_ADJUST) ? 1 : 0, bv.W, bv.x0, bv.obs,
                         bv.x_stride, bv.obs_stride, bv.N, vo_slab<float>(c, vo_off_p(c)), c->slab_seq);
  }
  else if (halves & 2)   // (after the candidates' half alone the buffer still holds the tracked positions the landmarks' half will read)
    hipLaunchKernelGGL(k_pipe_dense, dim3(B), dim3(PIPE_TPB), 0, c->stream, P, vo_slab<float>(c, vo_off_p(c)), c->slab_seq, 0);
  if (side) {
    // re-detection + spawn on the side stream behind promote / dense; adjustment + write-back on the main stream
    VO_HIP(c, hipEventRecord(c->ev_fork, c->stream));
    VO_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    c->stream = c->stream2;
    r = (stages & VO_PIPE_DETECT) ? vo_shi_tomasi_resident_counts(c, w->N, prm.mask_radius, &prm.st, w->d_dn + DN_PTS * B, w->d_dn + DN_ROOM * B) : VO_OK;
    if (r == VO_OK) pipe_launch_spawn(c, (stages & VO_PIPE_DETECT) ? 1 : 0, (stages & VO_PIPE_KEEP_FREE_LISTS) ? 0 : 1);
    c->stream = main_stream;
    if (r == VO_OK && (stages & VO_PIPE_ADJUST)) {
      // the live-slot counters are scoped to THIS enqueue: any other entry point that solves on the shared BA workspace afterwards
      // (vo_ba_solve_resident on a problem written through vo_ba_obs_device) must see every slot again
      vo_ba_set_live(c, (const int32_t*)w->tab[VO_PIPE_COUNTS] + C_NLM, PIPE_NCNT);
      r = vo_ba_enqueue_budget(c, &prm.ba, 0, prm.ba_budget);
      vo_ba_set_live(c, nullptr, 0);
    }
    if (r == VO_OK) PIPE_DISPATCH(k_pipe_writeback, P, (stages & VO_PIPE_ADJUST) ? 1 : 0, bv.pub, bv.pub_bytes, bv.x0, bv.x_stride, bv.W, w->d_rec);
    const hipError_t e1 = hipEventRecord(c->ev_join, c->stream2);
    const hipError_t e2 = hipStreamWaitEvent(c->stream, c->ev_join, 0);     // joined on every path: nothing is left running on the side stream
    if (r != VO_OK) return r;
    VO_HIP(c, e1); VO_HIP(c, e2);
  } else {
    if (stages & VO_PIPE_ADJUST) {
      vo_ba_set_live(c, (const int32_t*)w->tab[VO_PIPE_COUNTS] + C_NLM, PIPE_NCNT);
      r = vo_ba_enqueue_budget(c, &prm.ba, 0, prm.ba_budget);
      vo_ba_set_live(c, nullptr, 0);
      if (r != VO_OK) return r;
    }
    if (stages & VO_PIPE_DETECT) {
      r = vo_shi_tomasi_resident_counts(c, w->N, prm.mask_radius, &prm.st, w->d_dn + DN_PTS * B, w->d_dn + DN_ROOM * B);
      if (r != VO_OK) return r;
    }
    PIPE_DISPATCH(k_pipe_writeback, P, (stages & VO_PIPE_ADJUST) ? 1 : 0, bv.pub, bv.pub_bytes, bv.x0, bv.x_stride, bv.W, w->d_rec);
    pipe_launch_spawn(c, (stages & VO_PIPE_DETECT) ? 1 : 0, (stages & VO_PIPE_KEEP_FREE_LISTS) ? 0 : 1);
  }
  VO_HIP(c, hipGetLastError());
  const int slot = (int)(w->enq % VO_PIPE_INFLIGHT);
  VO_HIP(c, hipMemcpyAsync(w->h_rec + (size_t)slot * B, w->d_rec, sizeof(vo_pipe_record) * B, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipEventRecord(w->ev[slot], c->stream));
  w->enq++;
  return VO_OK;
}

extern "C" int32_t vo_pipe_fetch(vo_ctx* c, vo_pipe_record* rec) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->pipe && rec, VO_E_STATE, "vo_pipe_create first / null output");
  vo_pipe_ws* w = c->pipe;
  VO_CHECK(c, w->fetched < w->enq, VO_E_STATE, "no step in flight");
  VO_HIP(c, hipSetDevice(c->device));
  const int slot = (int)(w->fetched % VO_PIPE_INFLIGHT);
  VO_HIP(c, hipEventSynchronize(w->ev[slot]));
  memcpy(rec, w->h_rec + (size_t)slot * c->batch, sizeof(vo_pipe_record) * c->batch);
  w->fetched++;
  return VO_OK;
}
