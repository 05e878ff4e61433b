import sys
sys.path[:0]=['/root/repo/oracle','/root/repo/visual-odom-pipeline_amd']
import numpy as np, ba_oracle as bo
from vo_mi355x import VoContext, synthetic as syn
for W in (10, 16, 17, 20):
    s = syn.make_ba_scene(n_pts=300, n_slots=W, seed=5, visibility=0.85)
    with VoContext(64,64,max_pts=64) as c:
        c.ba_upload(s["K"], s["poses0"], s["points0"], s["obs"])
        pr = c.ba_probe(lam=1e-3)
    ne = bo.normal_equations(s["K"], s["poses0"], s["points0"], s["obs"])
    S, rhs, *_ = bo.schur_system(ne, 1e-3)
    D = np.abs(pr["S"]-S) / (np.abs(S)+1e-9*np.abs(S).max())
    bad = np.argwhere(D > 1e-6)
    print("W", W, "bad entries", len(bad), "rows", sorted(set(bad[:,0].tolist()))[:20], "cols", sorted(set(bad[:,1].tolist()))[:20], "rhs rel", np.linalg.norm(pr["rhs"]-rhs)/np.linalg.norm(rhs))
    A = np.abs(pr["S"]-S); i,j = np.unravel_index(np.argmax(A), A.shape)
    print("   frob rel", np.linalg.norm(pr["S"]-S)/np.linalg.norm(S), "max abs diff", A.max(), "at", (i,j), "S there", S[i,j], "gpu", pr["S"][i,j], "sym diff gpu", np.abs(pr["S"]-pr["S"].T).max())
