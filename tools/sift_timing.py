"""Timing of the SIFT call (host-synchronous: H2D of the image, scale space, refinement, host sort, descriptors, D2H)
on a 1241 x 376 frame, and of the whole bootstrap (2 x SIFT, matching, five-point pose, triangulation)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-odom-pipeline_amd"))
import numpy as np
from vo_mi355x import Extractor, VoContext, synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames, motions = syn.make_sequence(7, seed=5)
im0, im1 = frames[0], frames[6]
with VoContext(1241, 376, max_pts=2048, batch=B) as c:
    imgs = np.stack([im0] * B) if B > 1 else im0
    for _ in range(2):
        r = c.sift_detect_compute(imgs)
    t0 = time.perf_counter()
    for _ in range(10):
        r = c.sift_detect_compute(imgs)
    dt = (time.perf_counter() - t0) / 10
    n = len((r[0] if B > 1 else r)[0])
    print("B=%d sift_detect_compute 1241x376: %.2f ms per call (%.2f ms per image), %d keypoints" % (B, dt * 1e3, dt * 1e3 / B, n))
if B == 1:
    K = syn.KITTI_K
    with VoContext(1241, 376, max_pts=2048) as c:
        ext = Extractor(min_kp_dist=7, ctx=c)
        for rep in range(3):
            t0 = time.perf_counter()
            kp0 = ext.extract(im0, 0, detector='custom', describe=True)
            kp1 = ext.extract(im1, 1, detector='custom', describe=True)
            t1 = time.perf_counter()
            ms = ext.match_lists(kp0, kp1)
            t2 = time.perf_counter()
            k0 = [kp0[m.queryIdx] for m in ms]; k1 = [kp1[m.trainIdx] for m in ms]
            inl, H1 = ext.camera_pose(K, k0, k1, corr='2D-2D')
            t3 = time.perf_counter()
            lm, _, _ = ext.triangulate_nonlinear(K, np.eye(4), H1, [k0[i] for i in inl], [k1[i] for i in inl], 1, max_err_reproj=2.0)
            t4 = time.perf_counter()
        print("bootstrap through the drop-in Extractor: extract x2 %.1f ms, match_lists %.1f ms (%d matches), camera_pose %.1f ms (%d inliers), "
              "triangulate_nonlinear %.1f ms (%d landmarks); total %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, len(ms), (t3 - t2) * 1e3, len(inl),
                                                                              (t4 - t3) * 1e3, len(lm), (t4 - t0) * 1e3))
