"""Worker for tests/test_dist_cpu.py::test_sharded_ba_partition_world2 -- the landmark-sharded LM loop of config 5 with
gloo on CPU: every rank holds its landmark shard (vo_mi355x.sharding, the product's host logic), forms its part of the
reduced camera system with the numpy oracle, and the SAME two exchanges per iteration as csrc/vo_ba.hip run over
torch.distributed: all-reduce(sum) of the packet [E | r | Hpp | gp | cost | max|g_l| slot per rank], then of the 4
step statistics.  Every rank must reproduce the unsharded oracle solve (iteration sequence, cost, poses, points)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "visual-odom-pipeline_amd")]
import ba_oracle as bo  # noqa: E402
from vo_mi355x import sharding, synthetic as syn  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()


def allreduce(a):
    t = torch.from_numpy(np.ascontiguousarray(a, np.float64).copy())
    dist.all_reduce(t)
    return t.numpy()


V = 2                                      # shards per rank, as the batch dimension of a context would hold
S_total = world * V
s = syn.make_ba_scene(n_pts=301, n_slots=5, seed=4, visibility=0.9)
K, obs = s["K"], s["obs"]
W, N = obs.shape[:2]
_, po_s, pt_s, ob_s = sharding.shard_problem(K, s["poses0"], s["points0"], obs, S_total, first=rank * V, count=V)
poses = po_s[0].copy()
pts = pt_s.copy()                          # [V][Nloc][3]
lam, nu, ftol, xtol = 1e-4, 2.0, 1e-3, 1e-3
n = 6 * W
idx3, idx6 = np.arange(3), np.arange(6)
F = float(allreduce([sum(bo.cost(K, poses, pts[v], ob_s[v]) for v in range(V))])[0])
F0, n_acc, status, it = F, 0, 0, 0
for it in range(1, 31):
    nes = [bo.normal_equations(K, poses, pts[v], ob_s[v]) for v in range(V)]
    # ---- exchange 1: packet ----
    E, r, Hpp, gp, gmax = np.zeros((n, n)), np.zeros(n), np.zeros((W, 6, 6)), np.zeros((W, 6)), 0.0
    loc = []
    for ne in nes:
        Hd = ne["Hll"].copy()
        Hd[:, idx3, idx3] += lam * np.maximum(ne["Hll"][:, idx3, idx3], 1e-12)
        Minv = np.linalg.inv(Hd)
        B = ne["Hpl"].transpose(0, 2, 1, 3).reshape(n, -1, 3)
        z = np.einsum("ncd,nd->nc", Minv, ne["gl"])
        E += np.einsum("pnd,qnd->pq", np.einsum("pnc,ncd->pnd", B, Minv), B)
        r += np.einsum("pnc,nc->p", B, z)
        Hpp += ne["Hpp"]; gp += ne["gp"]
        gmax = max(gmax, np.abs(ne["gl"]).max())
        loc.append((Minv, B, z))
    slots = np.zeros(world); slots[rank] = gmax
    pk = allreduce(np.concatenate([E.ravel(), r, Hpp.ravel(), gp.ravel(), slots]))
    E, r = pk[:n * n].reshape(n, n), pk[n * n:n * n + n]
    Hpp = pk[n * n + n:n * n + n + 36 * W].reshape(W, 6, 6)
    gp = pk[n * n + n + 36 * W:n * n + n + 42 * W].reshape(W, 6)
    ginf = max(np.abs(gp).max(), pk[n * n + n + 42 * W:].max())
    if ginf < 1e-8:
        status = 1; it -= 1
        break
    Smat = -E
    Dp = np.maximum(Hpp[:, idx6, idx6], 1e-12)
    for i in range(W):
        blk = Hpp[i].copy(); blk[idx6, idx6] += lam * Dp[i]
        Smat[6 * i:6 * i + 6, 6 * i:6 * i + 6] += blk
    dp = np.linalg.solve(Smat, -gp.reshape(-1) + r)          # redundantly on every rank
    dpw = dp.reshape(W, 6)
    # ---- local back substitution, trial cost, exchange 2: statistics ----
    st4 = np.zeros(4)
    tpts = np.zeros_like(pts)
    for v, (ne, (Minv, B, z)) in enumerate(zip(nes, loc)):
        dl = -z - np.einsum("ncd,nd->nc", Minv, np.einsum("pnc,p->nc", B, dp))
        tpts[v] = pts[v] + dl
        Dl = np.maximum(ne["Hll"][:, idx3, idx3], 1e-12)
        st4 += [bo.cost(K, poses + dpw, tpts[v], ob_s[v]), lam * (Dl * dl * dl).sum() - (ne["gl"] * dl).sum(), (dl * dl).sum(),
                (pts[v] * pts[v]).sum()]
    Ft, predl, step2, x2 = allreduce(st4)
    pred = 0.5 * (predl + lam * (Dp * dpw * dpw).sum() - (gp * dpw).sum())
    step = np.sqrt(step2 + (dp * dp).sum())
    xn = np.sqrt(x2 + (poses * poses).sum())
    rho = (F - Ft) / pred if pred > 0 else -1.0
    if Ft < F and rho > 0:
        dF = F - Ft
        poses, pts, F = poses + dpw, tpts, Ft
        n_acc += 1
        lam = max(lam * max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3), 1e-3); nu = 2.0
        if dF < ftol * F:
            status = 2
            break
        if step < xtol * (xtol + xn):
            status = 3
            break
    else:
        if step < xtol * (xtol + xn):
            status = 3
            break
        lam *= nu; nu *= 2.0

ref = bo.solve(K, s["poses0"], s["points0"], obs, max_iters=30)
assert (it, n_acc, status) == (ref["iters"], ref["accepted"], ref["status"]), ((it, n_acc, status), ref["iters"], ref["accepted"], ref["status"])
assert abs(F - ref["cost"]) <= 1e-9 * ref["cost"] and abs(F0 - ref["cost0"]) <= 1e-9 * ref["cost0"]
assert np.abs(poses - ref["poses"]).max() <= 1e-8
# all-gather of the points once per adjust, then undo the dealing
out = [torch.zeros(pts.shape, dtype=torch.float64) for _ in range(world)]
dist.all_gather(out, torch.from_numpy(pts.copy()))
allpts = sharding.unshard_points(np.stack([o.numpy() for o in out]), N)
assert np.abs(allpts - ref["points"]).max() <= 1e-8
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
