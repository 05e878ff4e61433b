"""debug / timing probe: the reference's caller loop over the lazy drop-in classes on the GPU, per-frame sizes, session state and time"""
import copy, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "visual-odom-pipeline_amd"), ROOT]
import numpy as np
import bench
from vo_mi355x import BundleAdjuster, Extractor, VoContext, synthetic as syn

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
MAXP = int(os.environ.get("MAXP", "8192"))
lazy = (sys.argv[2] != "plain") if len(sys.argv) > 2 else True
sc = bench.pipe_scenes(1, 40, 4321)[0]
frames, K = sc["frames"], sc["K"]
with VoContext(1241, 376, max_pts=MAXP) as c:
    state, _ = syn.gt_bootstrap(c, sc, 0, bench.PIPE_T1)
    st = copy.deepcopy(state)
    ex = Extractor(min_kp_dist=7, ctx=c, lazy=lazy)
    ba = BundleAdjuster(verbosity=0, window_size=4, method='trf', xtol=1e-3, ftol=1e-3, ctx=c, max_iters=10)
    ex._im_prev = frames[bench.PIPE_T1]
    dead, dead_kp, t_step = [], [], 1
    for s in range(n_frames):
        im = frames[(bench.PIPE_T1 + 1 + s) % len(frames)]
        t0 = time.perf_counter(); t_step += 1
        tm = {}
        try:
            a = time.perf_counter(); st._candidates_kp = ex.extend_tracks(im, st._candidates_kp, max_bidir_error=np.inf); tm["et"] = time.perf_counter() - a
            a = time.perf_counter(); st._landmarks, st._landmarks_kp, ld, lkd = ex.extend_landmarks(im, st._landmarks, st._landmarks_kp, max_bidir_error=np.inf); tm["el"] = time.perf_counter() - a
            a = time.perf_counter(); dead += copy.deepcopy(ld); dead_kp += copy.deepcopy(lkd); ex._im_prev = im.copy(); tm["dc"] = time.perf_counter() - a
            a = time.perf_counter(); inl, Hk = ex.camera_pose(K, st._landmarks, st._landmarks_kp, corr='3D-2D', max_err_reproj=2.0); tm["cp"] = time.perf_counter() - a
            a = time.perf_counter()
            keep = inl if lazy else set(inl)
            lms, lkp = [], []
            for i in range(len(st._landmarks)):
                if i in keep:
                    lms.append(st._landmarks[i]); lkp.append(st._landmarks_kp[i])
                else:
                    dead.append(copy.deepcopy(st._landmarks[i])); dead_kp.append(copy.deepcopy(st._landmarks_kp[i]))
            st._landmarks, st._landmarks_kp = lms, lkp
            st._trajectory.append(t_step, Hk); tm["loop"] = time.perf_counter() - a
            a = time.perf_counter(); l_new, lk_new, st._candidates_kp = ex.triangulate_tracks(K, st._candidates_kp, st._trajectory, t_curr=t_step, min_track_length=3, min_bearing_angle=0.5, max_err_reproj=2.0, refine=True); tm["tt"] = time.perf_counter() - a
            st._landmarks_kp += lk_new; st._landmarks += l_new
            a = time.perf_counter(); st, dead, dead_kp = ba.adjust(st, dead, dead_kp, K, t_step); tm["ba"] = time.perf_counter() - a
            a = time.perf_counter(); st._candidates_kp += ex.extract(im, t_step, st._landmarks_kp + st._candidates_kp, detector='shi-tomasi', mask_radius=7, describe=False); tm["ex"] = time.perf_counter() - a
        except Exception:
            traceback.print_exc()
            print("FAILED at frame", s, "lm", len(st._landmarks), "cand", len(st._candidates_kp), "dead", len(dead))
            break
        sess = ex._lazy
        print("frame %2d  %6.2f ms  lm %4d cand %4d dead %4d  %s  %s" % (s, (time.perf_counter() - t0) * 1e3, len(st._landmarks), len(st._candidates_kp), len(dead),
              "plain" if sess is None else ("lazy fast=%d gathers=%d" % (sess.stats["fast"], sess.stats["gathers"])) + ("" if sess.alive else " DEAD: %s" % sess.reason),
              " ".join("%s %.2f" % (k, v * 1e3) for k, v in tm.items())), getattr(ex, "_lazy_error", ""), flush=True)
