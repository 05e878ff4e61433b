"""Array-native front end: numpy in, numpy out, one VoContext per (host thread, GPU).

Thin wrapper over the C ABI (include/vo_mi355x.h).  The drop-in classes in
extractor.py / bundle_adjuster.py are built on top of this.

Batched contexts: `VoContext(..., batch=B)` carries B independent sequences of identical shape in lockstep (one
launch serves all of them).  With B > 1 every array argument / result gains a leading dimension of length B; with
B == 1 (default) the arrays are exactly those of the single-sequence API.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BaParams, BaStats, EssParams, EssStats, KltParams, SiftKp, PnpParams, PnpStats, StParams, Tuning, VoError, as_c, ptr


class VoContext:
    # vo_tuning fields applied to every NEW context (a test module forces one kernel family for all the contexts its tests make this way; the
    # library itself reads no tuning from the environment)
    default_tuning = {}

    def __init__(self, width, height, max_pts=4096, device=0, max_level=3, win=31, batch=1):
        self._L = _lib.load()
        self._h = C.c_void_p()
        rc = self._L.vo_ctx_create_batched(device, width, height, max_pts, max_level, win, batch, C.byref(self._h))
        if rc != 0:
            msg = self._L.vo_last_error(None)
            raise VoError(rc, msg.decode() if msg else "vo_ctx_create failed")
        self.width, self.height, self.max_pts = width, height, max_pts
        self.max_level, self.win, self.device, self.batch = max_level, win, device, batch
        self._st_max_corners = 1000
        self._klt_levels = max_level + 1
        self.comm_ranks, self.comm_rank = 1, 0
        if self.default_tuning:
            self.set_tuning(**self.default_tuning)

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.vo_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != 0:
            msg = self._L.vo_last_error(self._h)
            raise VoError(rc, msg.decode() if msg else "")

    def sync(self):
        self._ck(self._L.vo_sync(self._h))

    def tuning(self):
        """-> dict of the vo_tuning fields in effect (0 = the library's rule)"""
        t = Tuning()
        self._ck(self._L.vo_get_tuning(self._h, C.byref(t)))
        return {k: getattr(t, k) for k in _lib.TUNING_FIELDS}

    def set_tuning(self, **fields):
        """force forms the library otherwise chooses by rule (include/vo_mi355x.h: vo_tuning; parity tests and A/B measurements).  Fields not
        named keep their value; `set_tuning(**dict.fromkeys(ctx.tuning(), 0))` restores the rules."""
        t = Tuning()
        self._ck(self._L.vo_get_tuning(self._h, C.byref(t)))
        for k, v in fields.items():
            if k not in _lib.TUNING_FIELDS:
                raise ValueError("set_tuning: unknown field %r" % k)
            setattr(t, k, int(v))
        self._ck(self._L.vo_set_tuning(self._h, C.byref(t)))

    # -- batch helpers --------------------------------------------------------------------------
    def _in(self, a, dtype, per_seq_shape):
        """-> C-contiguous [batch, *per_seq_shape] view of the caller's array (batch dim optional when batch == 1)"""
        a = np.asarray(a, dtype=dtype)
        want = (self.batch,) + tuple(per_seq_shape)
        if a.shape != want:
            if self.batch == 1:
                a = a.reshape(want)
            else:
                raise ValueError("expected array of shape %r, got %r" % (want, a.shape))
        return np.ascontiguousarray(a)

    def _out(self, a):
        return a[0] if self.batch == 1 else a

    # -- frames ---------------------------------------------------------------------------------
    def set_prefilter(self, d=5, sigma_color=1.5, sigma_space=1.5):
        """cv2.bilateralFilter(img, d, sigmaColor, sigmaSpace) on every frame entering the frame store (the reference
        loader's pre-filter, loader.py:16-20,86); d = 0 switches it off."""
        self._ck(self._L.vo_set_prefilter(self._h, int(d), float(sigma_color), float(sigma_space)))

    def bilateral(self, img, d=5, sigma_color=1.5, sigma_space=1.5):
        """cv2.bilateralFilter(img, d, sigmaColor, sigmaSpace) -> filtered uint8 image (Loader.getImage, reference
        loader.py:86).  Runs the fused pre-filter of the frame store and reads level 0 back, i.e. it PUSHES a frame: use a
        context of its own (vo_mi355x.Loader does), not the one an Extractor is tracking with."""
        self.set_prefilter(d, sigma_color, sigma_space)
        try:
            self.push_frame(img)
            return self.pyramid_read(1, 0)[0]
        finally:
            self.set_prefilter(0)

    def push_frame(self, img):
        img = np.asarray(img)
        if img.dtype != np.uint8:
            raise ValueError("push_frame: expected uint8")
        img = self._in(img, np.uint8, (self.height, self.width))
        self._ck(self._L.vo_frame_push(self._h, ptr(img, C.c_uint8), self.width))

    def upload_sequence(self, frames):
        """frames: [n_frames, h, w] (batch == 1) or [batch, n_frames, h, w]"""
        frames = np.asarray(frames, np.uint8)
        if frames.ndim == 3 and self.batch == 1:
            frames = frames[None]
        assert frames.ndim == 4 and frames.shape[0] == self.batch and frames.shape[2:] == (self.height, self.width)
        frames = np.ascontiguousarray(frames)
        self._ck(self._L.vo_seq_upload(self._h, ptr(frames, C.c_uint8), frames.shape[1]))

    def push_frame_resident(self, idx):
        self._ck(self._L.vo_frame_push_resident(self._h, int(idx)))

    def level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        self._ck(self._L.vo_pyramid_level_size(self._h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def pyramid_read(self, which, level, seq=0):
        """-> (img uint8 [h,w], deriv int16 [h,w,2]) of the prev (which=0) / cur (which=1) frame of sequence `seq`."""
        w, h = self.level_size(level)
        img = np.empty((h, w), np.uint8)
        der = np.empty((h, w, 2), np.int16)
        self._ck(self._L.vo_pyramid_read_seq(self._h, seq, which, level, ptr(img, C.c_uint8), ptr(der, C.c_int16)))
        return img, der

    # -- KLT ------------------------------------------------------------------------------------
    def klt_params(self, win=None, max_level=None, max_count=30, epsilon=0.03, min_eig_threshold=1e-4):
        p = KltParams()
        self._L.vo_klt_default_params(C.byref(p))
        p.win = self.win if win is None else win
        p.max_level = self.max_level if max_level is None else max_level
        p.max_count, p.epsilon, p.min_eig_threshold = max_count, epsilon, min_eig_threshold
        return p

    def _npts(self, p):
        p = np.asarray(p, np.float32)
        if self.batch == 1:
            p = p.reshape(1, -1, 2)
        if p.ndim != 3 or p.shape[0] != self.batch or p.shape[2] != 2:
            raise ValueError("points must have shape [batch, n, 2]")
        return np.ascontiguousarray(p), p.shape[1]

    def klt_track(self, p0, params=None, return_iters=False):
        """prev -> cur tracking.  p0 (n,2) float32 -> p1 (n,2) f32, status (n,) u8, err (n,) f32  [leading batch dim if batch > 1]"""
        p0, n = self._npts(p0)
        B = self.batch
        prm = params if params is not None else self.klt_params()
        p1 = np.zeros((B, n, 2), np.float32)
        st = np.zeros((B, n), np.uint8)
        err = np.zeros((B, n), np.float32)
        it = np.full((B, n, prm.max_level + 1), -1, np.int32)
        self._ck(self._L.vo_klt_track(self._h, ptr(p0, C.c_float), n, C.byref(prm), ptr(p1, C.c_float),
                                      ptr(st, C.c_uint8), ptr(err, C.c_float), ptr(it, C.c_int32)))
        if return_iters:
            return self._out(p1), self._out(st), self._out(err), self._out(it)
        return self._out(p1), self._out(st), self._out(err)

    def points_upload(self, p):
        p, n = self._npts(p)
        self._ck(self._L.vo_points_upload(self._h, ptr(p, C.c_float), n))

    def points_download(self, n, return_iters=False):
        B = self.batch
        p = np.zeros((B, n, 2), np.float32)
        st = np.zeros((B, n), np.uint8)
        err = np.zeros((B, n), np.float32)
        it = np.full((B, n, self._klt_levels), -1, np.int32) if return_iters else None
        self._ck(self._L.vo_points_download(self._h, ptr(p, C.c_float), ptr(st, C.c_uint8), ptr(err, C.c_float),
                                            ptr(it, C.c_int32), n))
        if return_iters:
            return self._out(p), self._out(st), self._out(err), self._out(it)
        return self._out(p), self._out(st), self._out(err)

    def klt_track_resident(self, n, params=None):
        prm = params if params is not None else self.klt_params()
        self._klt_levels = prm.max_level + 1
        self._ck(self._L.vo_klt_track_resident(self._h, n, C.byref(prm)))

    # -- Shi-Tomasi -----------------------------------------------------------------------------
    def st_params(self, max_corners=1000, quality_level=0.03, min_distance=7, block_size=31, use_harris=False, harris_k=0.04):
        p = StParams()
        self._L.vo_st_default_params(C.byref(p))
        p.max_corners, p.quality_level, p.min_distance, p.block_size = max_corners, quality_level, min_distance, block_size
        p.use_harris, p.harris_k = (1 if use_harris else 0), float(harris_k)
        return p

    def _corners(self, out, n_out):
        res = [out[b, :n_out[b]].copy() for b in range(self.batch)]
        return res[0] if self.batch == 1 else res

    def shi_tomasi(self, cur_pts=None, mask_radius=7, mask=None, params=None):
        """Re-detection on the CURRENT frame -> corners (m,2) float32 (integer-valued x,y) [list of B arrays if batch > 1]"""
        prm = params if params is not None else self.st_params()
        n_cur, pp = 0, None
        if cur_pts is not None and np.size(cur_pts):
            cur_pts, n_cur = self._npts(cur_pts)
            pp = ptr(cur_pts, C.c_float)
        mp = None
        if mask is not None:
            mask = self._in(mask, np.uint8, (self.height, self.width))
            mp = ptr(mask, C.c_uint8)
        mc = prm.max_corners if prm.max_corners > 0 else 4096
        out = np.zeros((self.batch, mc, 2), np.float32)
        n_out = np.zeros(self.batch, np.int32)
        self._ck(self._L.vo_shi_tomasi(self._h, pp, n_cur, int(mask_radius), mp, C.byref(prm), ptr(out, C.c_float),
                                       ptr(n_out, C.c_int32)))
        return self._corners(out, n_out)

    def shi_tomasi_resident(self, n_cur, mask_radius=7, params=None):
        prm = params if params is not None else self.st_params()
        self._st_max_corners = prm.max_corners
        self._ck(self._L.vo_shi_tomasi_resident(self._h, n_cur, int(mask_radius), C.byref(prm)))

    def shi_tomasi_fetch(self):
        mc = self._st_max_corners if self._st_max_corners > 0 else 4096
        out = np.zeros((self.batch, mc, 2), np.float32)
        n_out = np.zeros(self.batch, np.int32)
        self._ck(self._L.vo_shi_tomasi_fetch(self._h, ptr(out, C.c_float), ptr(n_out, C.c_int32)))
        return self._corners(out, n_out)

    def shi_tomasi_read(self):
        eig = np.empty((self.batch, self.height, self.width), np.float32)
        mask = np.empty((self.batch, self.height, self.width), np.uint8)
        nc = np.zeros(self.batch, np.int32)
        self._ck(self._L.vo_shi_tomasi_read(self._h, ptr(eig, C.c_float), ptr(mask, C.c_uint8), ptr(nc, C.c_int32)))
        if self.batch == 1:
            return eig[0], mask[0], int(nc[0])
        return eig, mask, nc

    # -- DLT ------------------------------------------------------------------------------------
    def _dlt_inputs(self, P0, P1, uv0, uv1, K, H0, H1):
        P0, P1 = self._in(P0, np.float32, (3, 4)), self._in(P1, np.float32, (3, 4))
        uv0, n = self._npts(uv0)
        uv1, n1 = self._npts(uv1)
        assert n == n1
        if K is not None:
            K, H0, H1 = self._in(K, np.float64, (3, 3)), self._in(H0, np.float64, (4, 4)), self._in(H1, np.float64, (4, 4))
        return P0, P1, uv0, uv1, n, K, H0, H1

    def triangulate(self, P0, P1, uv0, uv1, K=None, H0=None, H1=None):
        """cv2.triangulatePoints replacement.  -> X4 (4,n) f32 [, depth1 (n,) f64, reproj (n,) f64]"""
        P0, P1, uv0, uv1, n, K, H0, H1 = self._dlt_inputs(P0, P1, uv0, uv1, K, H0, H1)
        B, d = self.batch, C.c_double
        X4 = np.zeros((B, 4, n), np.float32)
        if K is None:
            self._ck(self._L.vo_triangulate_dlt(self._h, ptr(P0, C.c_float), ptr(P1, C.c_float), ptr(uv0, C.c_float),
                                                ptr(uv1, C.c_float), n, ptr(X4, C.c_float), None, None, None, None, None))
            return self._out(X4)
        depth, reproj = np.zeros((B, n)), np.zeros((B, n))
        self._ck(self._L.vo_triangulate_dlt(self._h, ptr(P0, C.c_float), ptr(P1, C.c_float), ptr(uv0, C.c_float),
                                            ptr(uv1, C.c_float), n, ptr(X4, C.c_float), ptr(K, d), ptr(H0, d), ptr(H1, d),
                                            ptr(depth, d), ptr(reproj, d)))
        return self._out(X4), self._out(depth), self._out(reproj)

    def dlt_upload(self, P0, P1, uv0, uv1, K=None, H0=None, H1=None):
        P0, P1, uv0, uv1, n, K, H0, H1 = self._dlt_inputs(P0, P1, uv0, uv1, K, H0, H1)
        self._dlt_n, self._dlt_stats = n, K is not None
        d = C.c_double
        self._ck(self._L.vo_dlt_upload(self._h, ptr(P0, C.c_float), ptr(P1, C.c_float), ptr(uv0, C.c_float),
                                       ptr(uv1, C.c_float), n, ptr(K, d), ptr(H0, d), ptr(H1, d)))

    def dlt_resident(self):
        self._ck(self._L.vo_dlt_resident(self._h))

    def dlt_fetch(self):
        n, B = self._dlt_n, self.batch
        X4, depth, reproj = np.zeros((B, 4, n), np.float32), np.zeros((B, n)), np.zeros((B, n))
        self._ck(self._L.vo_dlt_fetch(self._h, ptr(X4, C.c_float), ptr(depth, C.c_double), ptr(reproj, C.c_double)))
        return (self._out(X4), self._out(depth), self._out(reproj)) if self._dlt_stats else self._out(X4)

    # -- fused frame step ----------------------------------------------------------------------
    def set_graph_mode(self, on=True):
        self._ck(self._L.vo_set_graph_mode(self._h, 1 if on else 0))

    def set_side_stream(self, on=True):
        """re-detection + triangulation of the fused step on a side stream beside the bundle adjustment (True / 1, default) or in
        line (False / 0); 2 or "pipeline": three streams -- the bundle adjustment of frame t also runs beside the front end of
        frame t + 1 (two steps in flight)"""
        mode = 2 if on in (2, "pipeline") else (1 if on else 0)
        self._ck(self._L.vo_set_side_stream(self._h, mode))

    def step_layout(self):
        """-> {"layout": 0 | 1 | 2, "gate_groups": LM launch groups of frame t that precede the tracker launch of frame t + 1 (0: no gate),
        "reserved_cus": compute units the front-end stream leaves free} -- what vo_set_side_stream put into effect"""
        a, b, d = C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._L.vo_step_layout(self._h, C.byref(a), C.byref(b), C.byref(d)))
        return {"layout": a.value, "gate_groups": b.value, "reserved_cus": d.value}

    def frame_step_resident(self, frame_idx, n_pts, do_dlt=True, do_ba=True, do_st=True, mask_radius=7,
                            klt=None, st=None, ba=None):
        """enqueue one whole frame (pyramid, KLT, DLT, BA, re-detection, result copies); async"""
        klt = klt if klt is not None else self.klt_params()
        st = st if st is not None else self.st_params()
        ba = ba if ba is not None else self.ba_params()
        self._klt_levels = klt.max_level + 1
        self._st_max_corners = st.max_corners
        self._ck(self._L.vo_frame_step_resident(self._h, int(frame_idx), int(n_pts), int(do_dlt), int(do_ba), int(do_st),
                                                int(mask_radius), C.byref(klt), C.byref(st), C.byref(ba)))
        # what each step in flight will hand back (up to two are in flight, and they need not carry the same stages)
        if not hasattr(self, "_step_cfgs"):
            self._step_cfgs = []
        self._step_cfgs.append((n_pts, do_dlt, do_ba, do_st))

    def frame_step_host(self, frames, n_pts, do_dlt=True, do_ba=True, do_st=True, mask_radius=7, klt=None, st=None, ba=None):
        """the same step with this frame's images handed over by the host, as the reference's loop does (pipeline.py:98,171-172):
        frames = one uint8 [h, w] array per sequence (a list / tuple of `batch` arrays, rows contiguous, a common row stride) or one
        [batch, h, w] array.  The upload runs on a copy stream beside the previous step's bundle adjustment.  Arrays over pinned memory
        (`VoContext.host_alloc`) must stay untouched until `frame_fetch` has returned this step; the arrays are kept referenced until then."""
        ptrs, stride, frames = frames if isinstance(frames, tuple) and len(frames) == 3 and isinstance(frames[1], int) else self.host_frames(frames)
        klt = klt if klt is not None else self.klt_params()
        st = st if st is not None else self.st_params()
        ba = ba if ba is not None else self.ba_params()
        self._klt_levels = klt.max_level + 1
        self._st_max_corners = st.max_corners
        self._ck(self._L.vo_frame_step_host(self._h, ptrs, int(stride), int(n_pts), int(do_dlt), int(do_ba), int(do_st),
                                            int(mask_radius), C.byref(klt), C.byref(st), C.byref(ba)))
        if not hasattr(self, "_step_cfgs"):
            self._step_cfgs = []
        self._step_cfgs.append((n_pts, do_dlt, do_ba, do_st))
        if not hasattr(self, "_host_refs"):
            self._host_refs = []
        self._host_refs.append(frames)          # released by frame_fetch: DMA out of pinned arrays is asynchronous
        while len(self._host_refs) > 2:
            self._host_refs.pop(0)


    def host_frames(self, frames):
        """-> (pointer array, row stride, the arrays): the checked form of one step's images that `frame_step_host` also accepts directly (a loader
        that cycles through a fixed set of buffers builds it once per buffer)"""
        if isinstance(frames, np.ndarray):
            frames = [frames] if frames.ndim == 2 else list(frames)
        if len(frames) != self.batch:
            raise ValueError("frame_step_host: %d frames for a batch of %d" % (len(frames), self.batch))
        stride = None
        ptrs = (C.c_void_p * self.batch)()
        for b, f in enumerate(frames):
            if f.dtype != np.uint8 or f.shape != (self.height, self.width) or f.strides[1] != 1 or f.strides[0] < self.width:
                raise ValueError("frame_step_host: expected uint8 [%d, %d] images with contiguous rows" % (self.height, self.width))
            if stride is None:
                stride = f.strides[0]
            elif f.strides[0] != stride:
                raise ValueError("frame_step_host: the images of a step share one row stride")
            ptrs[b] = f.ctypes.data
        return ptrs, int(stride), frames

    @classmethod
    def host_alloc(cls, shape, dtype=np.uint8):
        """numpy array over page-locked host memory (vo_host_alloc): what a loader decodes frames into so that `frame_step_host` uploads them
        by DMA without a staging copy.  Freed when the array (and every view of it) is gone."""
        import weakref
        L = _lib.load()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        rc = L.vo_host_alloc(max(n, 1), C.byref(p))
        if rc != 0:
            raise VoError(rc, "vo_host_alloc(%d bytes) failed" % n)
        buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        weakref.finalize(buf, L.vo_host_free, C.c_void_p(p.value)).atexit = False      # (at interpreter exit the process's memory goes anyway; the HIP runtime may be gone first)
        return arr

    def frame_fetch(self):
        """wait for the enqueued frame and return its results as a dict of numpy arrays"""
        if getattr(self, "_step_cfgs", None):
            self._step_cfg = self._step_cfgs.pop(0)          # the OLDEST step not fetched yet (nothing in flight: the last one again)
        n_pts, do_dlt, do_ba, do_st = self._step_cfg
        B = self.batch
        p, stt, err = np.zeros((B, n_pts, 2), np.float32), np.zeros((B, n_pts), np.uint8), np.zeros((B, n_pts), np.float32)
        X4 = depth = reproj = poses = points = corners = None
        bs = (BaStats * B)()
        nc = np.zeros(B, np.int32)
        if do_dlt:
            n = self._dlt_n
            X4, depth, reproj = np.zeros((B, 4, n), np.float32), np.zeros((B, n)), np.zeros((B, n))
        if do_ba:
            W, N = self._ba_shape
            poses, points = np.zeros((B, W, 6)), np.zeros((B, N, 3))
        mc = self._st_max_corners if self._st_max_corners > 0 else 4096
        if do_st:
            corners = np.zeros((B, mc, 2), np.float32)
        d = C.c_double
        self._ck(self._L.vo_frame_fetch(self._h, n_pts, ptr(p, C.c_float), ptr(stt, C.c_uint8), ptr(err, C.c_float),
                                        ptr(X4, C.c_float), ptr(depth, d), ptr(reproj, d), ptr(poses, d), ptr(points, d),
                                        bs, ptr(corners, C.c_float), ptr(nc, C.c_int32) if do_st else None))
        out = dict(points2d=self._out(p), status=self._out(stt), err=self._out(err))
        if do_dlt:
            out.update(X4=self._out(X4), depth1=self._out(depth), reproj=self._out(reproj))
        if do_ba:
            stats = [self._stats(bs[b]) for b in range(B)]
            out.update(poses=self._out(poses), landmarks=self._out(points), ba_stats=stats[0] if B == 1 else stats)
        if do_st:
            out["corners"] = self._corners(corners, nc)
        return out

    # -- in-stream timing -----------------------------------------------------------------------
    PROF_FRAME, PROF_KLT, PROF_ST, PROF_DLT, PROF_BA = range(5)

    def profile_enable(self, regions=(0, 1, 2, 3, 4)):
        """regions: iterable of PROF_* ids to time with hipEvent pairs on the ctx stream; () switches timing off"""
        mask = 0
        for r in regions:
            mask |= 1 << r
        self._ck(self._L.vo_profile_enable(self._h, mask))

    def profile_read(self, region):
        t, n = C.c_double(0), C.c_int32(0)
        self._ck(self._L.vo_profile_read(self._h, region, C.byref(t), C.byref(n)))
        return t.value, n.value

    def debug_cycles(self, which):
        out = np.zeros(8, np.int64)
        self._ck(self._L.vo_debug_cycles(self._h, which, ptr(out, C.c_int64)))
        return out

    # -- BA -------------------------------------------------------------------------------------
    def ba_params(self, max_iters=50, ftol=1e-3, xtol=1e-3, gtol=1e-8, lambda0=1e-4, huber_delta=1.0, lambda_min=1e-3):
        p = BaParams()
        self._L.vo_ba_default_params(C.byref(p))
        p.max_iters, p.ftol, p.xtol, p.gtol, p.lambda0, p.huber_delta = max_iters, ftol, xtol, gtol, lambda0, huber_delta
        p.lambda_min = lambda_min
        return p

    @staticmethod
    def _stats(s):
        return dict(cost0=s.cost0, cost=s.cost, lam=s.lam, iters=s.iters, accepted=s.accepted, status=s.status,
                    n_obs=s.n_obs)

    def _ba_inputs(self, K, poses, points, obs):
        obs = np.asarray(obs, np.float64)
        if self.batch == 1 and obs.ndim == 3:
            obs = obs[None]
        assert obs.ndim == 4 and obs.shape[0] == self.batch and obs.shape[3] == 2
        W, N = obs.shape[1:3]
        return (self._in(K, np.float64, (3, 3)), self._in(poses, np.float64, (W, 6)), self._in(points, np.float64, (N, 3)),
                np.ascontiguousarray(obs), W, N)

    def ba_adjust(self, K, poses, points, obs, params=None):
        """poses (W,6) [rvec,tvec; slot 0 newest], points (N,3), obs (W,N,2) NaN = unobserved.
        -> poses (W,6), points (N,3), stats dict   [leading batch dim / list of dicts if batch > 1]"""
        K, poses, points, obs, W, N = self._ba_inputs(K, poses, points, obs)
        B = self.batch
        prm = params if params is not None else self.ba_params()
        po, pt, st = np.zeros((B, W, 6)), np.zeros((B, N, 3)), (BaStats * B)()
        d = C.c_double
        self._ck(self._L.vo_ba_adjust(self._h, ptr(K, d), ptr(poses, d), ptr(points, d), ptr(obs, d), W, N, C.byref(prm),
                                      ptr(po, d), ptr(pt, d), st))
        self._ba_shape = (W, N)
        stats = [self._stats(st[b]) for b in range(B)]
        return self._out(po), self._out(pt), stats[0] if B == 1 else stats

    def ba_upload(self, K, poses, points, obs):
        K, poses, points, obs, W, N = self._ba_inputs(K, poses, points, obs)
        self._ba_shape = (W, N)
        d = C.c_double
        self._ck(self._L.vo_ba_upload(self._h, ptr(K, d), ptr(poses, d), ptr(points, d), ptr(obs, d), W, N))

    def ba_upload_bank(self, K, poses, points, obs):
        """a bank of resident problems of one shape: poses [n, B, W, 6], points [n, B, N, 3], obs [n, B, W, N, 2] (B = batch);
        ba_select(k) picks the one the next resident solve / frame step works on (no upload, no launch)"""
        obs = np.ascontiguousarray(obs, np.float64)
        n, B, W, N = obs.shape[:4]
        assert B == self.batch and obs.shape[4] == 2
        poses = np.ascontiguousarray(poses, np.float64).reshape(n, B, W, 6)
        points = np.ascontiguousarray(points, np.float64).reshape(n, B, N, 3)
        K = self._in(K, np.float64, (3, 3))
        d = C.c_double
        self._ck(self._L.vo_ba_upload_bank(self._h, ptr(K, d), ptr(poses, d), ptr(points, d), ptr(obs, d), W, N, n))
        self._ba_shape = (W, N)

    def ba_select(self, k):
        self._ck(self._L.vo_ba_select_problem(self._h, int(k)))

    def ba_solve_resident(self, params=None):
        prm = params if params is not None else self.ba_params()
        self._ck(self._L.vo_ba_solve_resident(self._h, C.byref(prm)))

    def ba_fetch(self):
        W, N = self._ba_shape
        B = self.batch
        po, pt, st = np.zeros((B, W, 6)), np.zeros((B, N, 3)), (BaStats * B)()
        self._ck(self._L.vo_ba_fetch(self._h, ptr(po, C.c_double), ptr(pt, C.c_double), st))
        stats = [self._stats(st[b]) for b in range(B)]
        return self._out(po), self._out(pt), stats[0] if B == 1 else stats

    def ba_fetch_after(self, solve, params):
        """solve(params) then ba_fetch() -- a resident solve whose result is wanted at once"""
        solve(params)
        return self.ba_fetch()

    # -- 3D-2D pose -----------------------------------------------------------------------------
    def pnp_ransac(self, K, pts3d, pts2d, reproj_err=2.0, confidence=0.9999, max_iters=1000000, seed=0):
        """RANSAC P3P + refinement.  pts3d (n,3), pts2d (n,2) [leading batch dim if batch > 1]
        -> rvec (3,), tvec (3,), inlier indices (ascending), stats dict   [lists / arrays over the batch]"""
        B = self.batch
        p3 = np.ascontiguousarray(pts3d, np.float32).reshape(B, -1, 3)
        p2 = np.ascontiguousarray(pts2d, np.float32).reshape(B, -1, 2)
        n = p3.shape[1]
        assert p2.shape[1] == n
        Kc = self._in(K, np.float64, (3, 3))
        prm = PnpParams()
        self._L.vo_pnp_default_params(C.byref(prm))
        prm.reproj_err, prm.confidence, prm.max_iters, prm.seed = reproj_err, confidence, int(max_iters), int(seed)
        rv, tv = np.zeros((B, 3)), np.zeros((B, 3))
        mask = np.zeros((B, n), np.uint8)
        st = (PnpStats * B)()
        self._ck(self._L.vo_pnp_ransac(self._h, ptr(Kc, C.c_double), ptr(p3, C.c_float), ptr(p2, C.c_float), n, C.byref(prm),
                                       ptr(rv, C.c_double), ptr(tv, C.c_double), ptr(mask, C.c_uint8), st))
        stats = [dict(cost=s.cost, n_inliers=s.n_inliers, hypotheses=s.hypotheses, best=s.best, status=s.status) for s in st]
        inl = [np.nonzero(mask[b])[0] for b in range(B)]
        if B == 1:
            return rv[0], tv[0], inl[0], stats[0]
        return rv, tv, inl, stats

    # -- SIFT ---------------------------------------------------------------------------------------
    def sift_detect_compute(self, img, mask=None, nfeatures=1000, max_out=None):
        """cv2.SIFT_create(nfeatures).detect(img, mask) + compute.  img (h, w) u8 [leading batch dim if batch > 1]
        -> keypoints (n, 6) float64 [x, y, size, angle, response, octave], descriptors (n, 128) float32   [lists over the batch]"""
        B = self.batch
        im = np.ascontiguousarray(img, np.uint8).reshape(B, self.height, self.width)
        mk = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(B, self.height, self.width)
        cap = int(max_out) if max_out else (2 * nfeatures + 64 if nfeatures > 0 else 1 << 16)
        kps = (SiftKp * (B * cap))()
        desc = np.zeros((B, cap, 128), np.float32)
        n_out = np.zeros(B, np.int32)
        self._ck(self._L.vo_sift_detect_compute(self._h, ptr(im, C.c_uint8), self.width, None if mk is None else ptr(mk, C.c_uint8),
                                                int(nfeatures), cap, kps, ptr(desc, C.c_float), ptr(n_out, C.c_int32)))
        raw = np.frombuffer(kps, dtype=np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                                                 ("octave", "<i4")])).reshape(B, cap)
        outs = []
        for b in range(B):
            r = raw[b, :n_out[b]]
            k = np.stack([r["x"], r["y"], r["size"], r["angle"], r["response"], r["octave"]], 1).astype(np.float64).reshape(-1, 6)
            outs.append((k, desc[b, :n_out[b]].copy()))
        return outs[0] if B == 1 else outs

    # -- descriptor matching ----------------------------------------------------------------------
    def match_knn2(self, desc1, desc2):
        """two nearest train descriptors (L2) of every query descriptor.  desc1 (n1,dim), desc2 (n2,dim) f32
        [leading batch dim if batch > 1] -> idx (n1,2) int32 (-1 = none), dist (n1,2) float32"""
        B = self.batch
        dim = np.shape(desc1)[-1]
        d1 = np.ascontiguousarray(desc1, np.float32).reshape(B, -1, dim)
        d2 = np.ascontiguousarray(desc2, np.float32).reshape(B, -1, dim)
        n1, n2 = d1.shape[1], d2.shape[1]
        idx = np.zeros((B, n1, 2), np.int32); dist = np.zeros((B, n1, 2), np.float32)
        self._ck(self._L.vo_match_knn2(self._h, ptr(d1, C.c_float), n1, ptr(d2, C.c_float), n2, dim, ptr(idx, C.c_int32),
                                       ptr(dist, C.c_float)))
        return (idx[0], dist[0]) if B == 1 else (idx, dist)

    # -- 2D-2D bootstrap pose ---------------------------------------------------------------------
    def essential_ransac(self, K, pts1, pts2, threshold=1.0, prob=0.9999, max_iters=1000, seed=0, distance_thresh=50.0):
        """findEssentialMat(RANSAC) + recoverPose.  pts1, pts2 (n,2) pixels [leading batch dim if batch > 1]
        -> E (3,3) unit norm, R (3,3), t (3,) with x2 ~ R x1 + t, inlier indices (ascending), stats dict"""
        B = self.batch
        p1 = np.ascontiguousarray(pts1, np.float32).reshape(B, -1, 2)
        p2 = np.ascontiguousarray(pts2, np.float32).reshape(B, -1, 2)
        n = p1.shape[1]
        assert p2.shape[1] == n
        Kc = self._in(K, np.float64, (3, 3))
        prm = EssParams()
        self._L.vo_essential_default_params(C.byref(prm))
        prm.threshold, prm.prob, prm.distance_thresh, prm.max_iters, prm.seed = threshold, prob, distance_thresh, int(max_iters), int(seed)
        E, R, t = np.zeros((B, 3, 3)), np.zeros((B, 3, 3)), np.zeros((B, 3))
        mask = np.zeros((B, n), np.uint8)
        st = (EssStats * B)()
        self._ck(self._L.vo_essential_ransac(self._h, ptr(Kc, C.c_double), ptr(p1, C.c_float), ptr(p2, C.c_float), n, C.byref(prm),
                                             ptr(E, C.c_double), ptr(R, C.c_double), ptr(t, C.c_double), ptr(mask, C.c_uint8), st))
        stats = [dict(n_inliers=s.n_inliers, n_good=s.n_good, hypotheses=s.hypotheses, best=s.best, status=s.status) for s in st]
        inl = [np.nonzero(mask[b])[0] for b in range(B)]
        if B == 1:
            return E[0], R[0], t[0], inl[0], stats[0]
        return E, R, t, inl, stats

    def pnp_params(self, reproj_err=2.0, confidence=0.9999, max_iters=1000000, seed=0):
        prm = PnpParams()
        self._L.vo_pnp_default_params(C.byref(prm))
        prm.reproj_err, prm.confidence, prm.max_iters, prm.seed = reproj_err, confidence, int(max_iters), int(seed)
        return prm

    def pnp_upload(self, K, pts3d, pts2d):
        B = self.batch
        p3 = np.ascontiguousarray(pts3d, np.float32).reshape(B, -1, 3)
        p2 = np.ascontiguousarray(pts2d, np.float32).reshape(B, -1, 2)
        self._pnp_n = p3.shape[1]
        Kc = self._in(K, np.float64, (3, 3))
        self._ck(self._L.vo_pnp_upload(self._h, ptr(Kc, C.c_double), ptr(p3, C.c_float), ptr(p2, C.c_float), self._pnp_n))

    def pnp_solve_resident(self, params=None, blind_batches=2):
        prm = params if params is not None else self.pnp_params()
        self._ck(self._L.vo_pnp_solve_resident(self._h, C.byref(prm), int(blind_batches)))

    def pnp_fetch(self):
        B, n = self.batch, self._pnp_n
        rv, tv, mask, st = np.zeros((B, 3)), np.zeros((B, 3)), np.zeros((B, n), np.uint8), (PnpStats * B)()
        self._ck(self._L.vo_pnp_fetch(self._h, ptr(rv, C.c_double), ptr(tv, C.c_double), ptr(mask, C.c_uint8), st))
        stats = [dict(cost=s.cost, n_inliers=s.n_inliers, hypotheses=s.hypotheses, best=s.best, status=s.status) for s in st]
        inl = [np.nonzero(mask[b])[0] for b in range(B)]
        if B == 1:
            return rv[0], tv[0], inl[0], stats[0]
        return rv, tv, inl, stats

    # -- device-resident track table ------------------------------------------------------------
    def tracks_seed(self, pts, t=0):
        """Initial tracks born at frame t; pts (n,2) [(B,n,2)]."""
        pts, n = self._npts(pts)
        self._ck(self._L.vo_tracks_seed(self._h, ptr(pts, C.c_float), n, int(t)))

    def tracks_track(self, t, params=None):
        """KLT prev -> cur of every live track + the reference's pruning / bookkeeping (async)."""
        prm = params if params is not None else self.klt_params()
        self._ck(self._L.vo_tracks_track(self._h, int(t), C.byref(prm)))

    def tracks_detect(self, t, mask_radius=7, params=None, max_new=1000):
        """Shi-Tomasi re-detection around the live tracks; corners become tracks born at t (async)."""
        prm = params if params is not None else self.st_params()
        self._ck(self._L.vo_tracks_detect(self._h, int(t), int(mask_radius), C.byref(prm), int(max_new)))

    def tracks_read(self):
        """-> list (one dict per sequence; a dict if batch == 1) of uv, uv_first (n,2) f32, t_first, t_total, tag (n,),
        dead_tag (n_dead,)."""
        B, cap = self.batch, self.max_pts
        n, nd = np.zeros(B, np.int32), np.zeros(B, np.int32)
        uv, uf = np.zeros((B, cap, 2), np.float32), np.zeros((B, cap, 2), np.float32)
        tf, tt, tag, dead = (np.zeros((B, cap), np.int32) for _ in range(4))
        i, f = C.c_int32, C.c_float
        self._ck(self._L.vo_tracks_read(self._h, ptr(n, i), ptr(uv, f), ptr(uf, f), ptr(tf, i), ptr(tt, i), ptr(tag, i),
                                        ptr(nd, i), ptr(dead, i)))
        out = [dict(uv=uv[b, :n[b]].copy(), uv_first=uf[b, :n[b]].copy(), t_first=tf[b, :n[b]].copy(),
                    t_total=tt[b, :n[b]].copy(), tag=tag[b, :n[b]].copy(), dead_tag=dead[b, :nd[b]].copy()) for b in range(B)]
        return out[0] if B == 1 else out

    def tracks_counts(self):
        """-> (n_live, n_dead) per sequence [(B,) int32 each; ints if batch == 1]: the synchronous read-back of the counters only
        (vo_tracks_read with NULL tables) -- what a per-frame loop needs when the tables stay on the device"""
        B = self.batch
        n, nd = np.zeros(B, np.int32), np.zeros(B, np.int32)
        i = C.c_int32
        self._ck(self._L.vo_tracks_read(self._h, ptr(n, i), None, None, None, None, None, ptr(nd, i), None))
        return (int(n[0]), int(nd[0])) if B == 1 else (n, nd)

    def tracks_obs(self, t_now, window):
        """BA observation table of the live tracks -> (B, window, max_pts, 2) f64 [(window, max_pts, 2) if batch == 1]."""
        obs = np.zeros((self.batch, window, self.max_pts, 2))
        self._ck(self._L.vo_tracks_obs(self._h, int(t_now), int(window), ptr(obs, C.c_double)))
        return obs[0] if self.batch == 1 else obs

    def ba_obs_from_tracks(self, t_now):
        """Fill the observation table of the resident BA problem (N = max_pts, W slots) from the track ring (async)."""
        self._ck(self._L.vo_ba_obs_from_tracks(self._h, int(t_now)))

    # -- landmark-sharded BA of one problem (config 5) -------------------------------------------
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id (rank 0 makes it, every rank passes it to comm_init)."""
        L = _lib.load()
        uid = np.zeros(128, np.uint8)
        rc = L.vo_comm_unique_id(ptr(uid, C.c_uint8))
        if rc != 0:
            raise RuntimeError("vo_comm_unique_id failed (%d): librccl not loadable?" % rc)
        return uid

    def comm_init(self, n_ranks, rank, uid):
        uid = np.ascontiguousarray(uid, np.uint8)
        assert uid.size == 128
        self._ck(self._L.vo_comm_init(self._h, int(n_ranks), int(rank), ptr(uid, C.c_uint8)))
        self.comm_ranks, self.comm_rank = int(n_ranks), int(rank)

    def comm_destroy(self):
        self._ck(self._L.vo_comm_destroy(self._h))
        self.comm_ranks, self.comm_rank = 1, 0

    def ba_set_sharded(self, on=True):
        """The `batch` problems of ba_upload / ba_adjust (x the ranks of the communicator) are landmark shards of
        ONE problem (see sharding.shard_problem)."""
        self._ck(self._L.vo_ba_set_sharded(self._h, 1 if on else 0))

    def ba_gather_points(self):
        """-> points of every shard of every rank, (n_ranks, batch, N, 3)."""
        W, N = self._ba_shape
        R = getattr(self, "comm_ranks", 1)
        out = np.zeros((R, self.batch, N, 3))
        self._ck(self._L.vo_ba_gather_points(self._h, ptr(out, C.c_double)))
        return out

    def ba_probe(self, lam=1e-4, huber_delta=1.0):
        """Parity probe of problem 0 at the uploaded x0: residuals, normal equations, reduced system, one LM step."""
        W, N = self._ba_shape
        res = np.zeros(W * N)
        n_obs = C.c_int32(0)
        cost = C.c_double(0)
        Hpp, gp = np.zeros((W, 6, 6)), np.zeros((W, 6))
        Hll, gl = np.zeros((N, 3, 3)), np.zeros((N, 3))
        S, rhs = np.zeros((6 * W, 6 * W)), np.zeros(6 * W)
        dp, dl = np.zeros((W, 6)), np.zeros((N, 3))
        d = C.c_double
        self._ck(self._L.vo_ba_probe(self._h, lam, huber_delta, ptr(res, d), C.byref(n_obs), C.byref(cost),
                                     ptr(Hpp, d), ptr(gp, d), ptr(Hll, d), ptr(gl, d), ptr(S, d), ptr(rhs, d),
                                     ptr(dp, d), ptr(dl, d)))
        return dict(residual=res[:n_obs.value].copy(), cost=cost.value, Hpp=Hpp, gp=gp, Hll=Hll, gl=gl, S=S, rhs=rhs,
                    dposes=dp, dpoints=dl)
