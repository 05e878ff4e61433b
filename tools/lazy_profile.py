"""cProfile of the reference's caller loop over the lazy drop-in classes on the GPU (frames 2..N: all six calls on the fast path)"""
import copy, cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "visual-odom-pipeline_amd"), ROOT]
import numpy as np
import bench
from vo_mi355x import BundleAdjuster, Extractor, VoContext, synthetic as syn

sc = bench.pipe_scenes(1, 40, 4321)[0]
frames, K = sc["frames"], sc["K"]
with VoContext(1241, 376, max_pts=4096) as c:
    state, _ = syn.gt_bootstrap(c, sc, 0, bench.PIPE_T1)
    st = copy.deepcopy(state)
    ex = Extractor(min_kp_dist=7, ctx=c, lazy=True)
    ba = BundleAdjuster(verbosity=0, window_size=4, method='trf', xtol=1e-3, ftol=1e-3, ctx=c, max_iters=10)
    ex._im_prev = frames[bench.PIPE_T1]
    box = dict(st=st, dead=[], dead_kp=[], t=1)

    def step(s):
        st, dead, dead_kp = box["st"], box["dead"], box["dead_kp"]
        im = frames[(bench.PIPE_T1 + 1 + s) % len(frames)]
        box["t"] += 1
        t_step = box["t"]
        st._candidates_kp = ex.extend_tracks(im, st._candidates_kp, max_bidir_error=np.inf)
        st._landmarks, st._landmarks_kp, ld, lkd = ex.extend_landmarks(im, st._landmarks, st._landmarks_kp, max_bidir_error=np.inf)
        dead += copy.deepcopy(ld); dead_kp += copy.deepcopy(lkd)
        ex._im_prev = im.copy()
        inl, Hk = ex.camera_pose(K, st._landmarks, st._landmarks_kp, corr='3D-2D', max_err_reproj=2.0)
        lms, lkp = [], []
        for i in range(len(st._landmarks)):
            if i in inl:
                lms.append(st._landmarks[i]); lkp.append(st._landmarks_kp[i])
            else:
                dead.append(copy.deepcopy(st._landmarks[i])); dead_kp.append(copy.deepcopy(st._landmarks_kp[i]))
        st._landmarks, st._landmarks_kp = lms, lkp
        st._trajectory.append(t_step, Hk)
        l_new, lk_new, st._candidates_kp = ex.triangulate_tracks(K, st._candidates_kp, st._trajectory, t_curr=t_step, min_track_length=3, min_bearing_angle=0.5, max_err_reproj=2.0, refine=True)
        st._landmarks_kp += lk_new; st._landmarks += l_new
        box["st"], box["dead"], box["dead_kp"] = ba.adjust(st, dead, dead_kp, K, t_step)
        st = box["st"]
        st._candidates_kp += ex.extract(im, t_step, st._landmarks_kp + st._candidates_kp, detector='shi-tomasi', mask_radius=7, describe=False)

    for s in range(6):
        step(s)
    t0 = time.perf_counter()
    for s in range(6, 12):
        step(s)
    print("unprofiled: %.3f ms per frame" % ((time.perf_counter() - t0) / 6 * 1e3), ex._lazy.stats, ex._lazy.alive)
    pr = cProfile.Profile()
    pr.enable()
    for s in range(12, 16):
        step(s)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
