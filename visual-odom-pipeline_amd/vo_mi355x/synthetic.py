"""Synthetic inputs for tests and bench.py.

There are no datasets in the image and no network, so every measurement runs on
seeded synthetic data of the BASELINE shapes:

* images: band-limited noise texture warped by a known similarity motion per
  frame (gives KLT an analytic ground truth),
* BA scene: KITTI-like camera, forward motion, noisy observations.
"""
import numpy as np

KITTI_K = np.array([[718.856, 0.0, 607.1928], [0.0, 718.856, 185.2157], [0.0, 0.0, 1.0]])


def _gauss_kernel(sigma):
    r = int(np.ceil(3 * sigma))
    x = np.arange(-r, r + 1, dtype=np.float64)
    k = np.exp(-0.5 * (x / sigma) ** 2)
    return (k / k.sum()).astype(np.float32)


def _blur(img, sigma):
    k = _gauss_kernel(sigma)
    r = len(k) // 2
    p = np.pad(img, ((0, 0), (r, r)), mode="reflect")
    out = np.zeros_like(img)
    for i, kv in enumerate(k):
        out += kv * p[:, i:i + img.shape[1]]
    p = np.pad(out, ((r, r), (0, 0)), mode="reflect")
    out2 = np.zeros_like(img)
    for i, kv in enumerate(k):
        out2 += kv * p[i:i + img.shape[0], :]
    return out2


def make_texture(h, w, seed=1234):
    """float32 texture, mean 128 / std 40, two octaves (sigma 2 px + 0.5 * sigma 6 px)."""
    rng = np.random.default_rng(seed)
    n = rng.standard_normal((h, w)).astype(np.float32)
    a = _blur(n, 2.0)
    b = _blur(n, 6.0)
    t = a / a.std() + 0.5 * b / b.std()
    t = (t - t.mean()) / t.std()
    return (128.0 + 40.0 * t).astype(np.float32)


def frame_motion(t, w, h):
    """2x3 affine A_t mapping frame-0 pixel coordinates to frame-t coordinates:
    translation (2.0 t, 0.7 t) px, rotation 0.002 t rad about the image centre,
    scale 1 + 0.003 t."""
    c = np.array([(w - 1) * 0.5, (h - 1) * 0.5])
    ang, s = 0.002 * t, 1.0 + 0.003 * t
    R = s * np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    tr = np.array([2.0 * t, 0.7 * t])
    A = np.zeros((2, 3))
    A[:, :2] = R
    A[:, 2] = c - R @ c + tr
    return A


def warp_points(A, pts):
    pts = np.asarray(pts, np.float64).reshape(-1, 2)
    return pts @ A[:, :2].T + A[:, 2]


def render_frame(tex, A, w, h, margin):
    """Sample the (larger) texture at the inverse-warped pixel grid, bilinear.
    tex covers frame-0 coordinates [-margin, w+margin) x [-margin, h+margin)."""
    Ainv = np.linalg.inv(np.vstack([A, [0, 0, 1]]))[:2]
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    sx = Ainv[0, 0] * xs + Ainv[0, 1] * ys + Ainv[0, 2] + margin
    sy = Ainv[1, 0] * xs + Ainv[1, 1] * ys + Ainv[1, 2] + margin
    x0 = np.floor(sx).astype(np.int64)
    y0 = np.floor(sy).astype(np.int64)
    fx = (sx - x0).astype(np.float32)
    fy = (sy - y0).astype(np.float32)
    x0 = np.clip(x0, 0, tex.shape[1] - 2)
    y0 = np.clip(y0, 0, tex.shape[0] - 2)
    v = (tex[y0, x0] * (1 - fx) * (1 - fy) + tex[y0, x0 + 1] * fx * (1 - fy) +
         tex[y0 + 1, x0] * (1 - fx) * fy + tex[y0 + 1, x0 + 1] * fx * fy)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def frame_motion_periodic(t, w, h, period):
    """like frame_motion, on a closed curve: the similarity sways with period `period` frames (frame `period` = frame 0), at most
    ~2.5 px, 0.0025 rad and 0.4 % of scale per frame for a period of 100 -- a sequence that can be played in a loop of any length"""
    c = np.array([(w - 1) * 0.5, (h - 1) * 0.5])
    ph = 2 * np.pi * t / period
    ang, s = 0.04 * np.sin(ph), 1.0 + 0.06 * np.sin(ph + 0.7)
    R = s * np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    tr = np.array([40.0 * np.sin(ph), 14.0 * np.sin(2 * ph)])
    A = np.zeros((2, 3))
    A[:, :2] = R
    A[:, 2] = c - R @ c + tr
    return A


def make_sequence(n_frames, w=1241, h=376, seed=1234, margin=96, periodic=False, n_render=None):
    """-> (frames uint8 [n, h, w], motions [n, 2, 3]); periodic: frame n_frames would equal frame 0 (play it in a loop);
    n_render: only the first n_render frames of the n_frames-long motion are rendered"""
    tex = make_texture(h + 2 * margin, w + 2 * margin, seed)
    n_out = n_frames if n_render is None else min(n_render, n_frames)
    frames = np.empty((n_out, h, w), np.uint8)
    motions = np.empty((n_out, 2, 3))
    for t in range(n_out):
        A = frame_motion_periodic(t, w, h, n_frames) if periodic else frame_motion(t, w, h)
        motions[t] = A
        frames[t] = render_frame(tex, A, w, h, margin)
    return frames, motions


def grid_points(n, w, h, margin=24, seed=7):
    """n jittered-grid keypoints, float32 (n, 2), >= margin px from the border."""
    rng = np.random.default_rng(seed)
    aspect = (w - 2 * margin) / (h - 2 * margin)
    ny = max(1, int(np.sqrt(n / aspect)))
    nx = int(np.ceil(n / ny))
    xs = np.linspace(margin, w - 1 - margin, nx)
    ys = np.linspace(margin, h - 1 - margin, ny)
    g = np.stack(np.meshgrid(xs, ys), -1).reshape(-1, 2)[:n]
    g = g + rng.uniform(-2.0, 2.0, g.shape)
    return g.astype(np.float32)


# ----------------------------------------------------------------------------
# BA scene
# ----------------------------------------------------------------------------
def rodrigues(r):
    r = np.asarray(r, np.float64).reshape(3)
    th = np.linalg.norm(r)
    if th < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def make_ba_scene(n_pts=2000, n_slots=10, K=KITTI_K, seed=0, obs_noise=0.3, pt_noise=0.3,
                  pose_noise=0.02, visibility=1.0, width=1241, height=376):
    """Sliding-window BA problem in the layout the C-ABI takes.

    Slot 0 is the NEWEST frame (the reference packs poses newest-first,
    /root/reference/src/bundle_adjuster/bundle_adjuster.py:169-176).
    Returns dict(K, poses0 [W,6] (rvec,tvec), points0 [N,3], obs [W,N,2] (NaN =
    not observed), poses_gt, points_gt)."""
    rng = np.random.default_rng(seed)
    W, N = n_slots, n_pts
    pts = np.stack([rng.uniform(-15, 15, N), rng.uniform(-3, 3, N), rng.uniform(12, 60, N)], 1)
    poses_gt = np.zeros((W, 6))
    for i in range(W):
        f = W - 1 - i  # frame index in time; slot 0 = newest
        yaw = 0.01 * f
        Rwc = rodrigues([0, yaw, 0])  # camera orientation in world
        cam_c = np.array([0.0, 0.0, 0.8 * f])
        R = Rwc.T  # world -> camera
        t = -R @ cam_c
        # rvec of R
        th = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
        if th < 1e-12:
            rv = np.zeros(3)
        else:
            rv = th / (2 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        poses_gt[i, :3], poses_gt[i, 3:] = rv, t
    obs = np.full((W, N, 2), np.nan)
    for i in range(W):
        R = rodrigues(poses_gt[i, :3])
        Xc = pts @ R.T + poses_gt[i, 3:]
        p = Xc @ K.T
        uv = p[:, :2] / p[:, 2:3]
        uv += rng.normal(0, obs_noise, uv.shape)
        vis = Xc[:, 2] > 0.5
        if visibility < 1.0:
            vis &= rng.uniform(size=N) < visibility
        obs[i, vis] = uv[vis]
    points0 = pts + rng.normal(0, pt_noise, pts.shape)
    poses0 = poses_gt.copy()
    poses0[:-2, 3:] += rng.normal(0, pose_noise, (W - 2, 3)) if W > 2 else 0.0
    return dict(K=np.array(K, np.float64), poses0=poses0, points0=points0, obs=obs,
                poses_gt=poses_gt, points_gt=pts)


def make_two_plane_sequence(n_frames, w=640, h=480, f=500.0, seed=2024, z_bg=10.0, z_fg=6.5, roll=0.004,
                            step=(0.10, 0.03, -0.06), margin=128):
    """A rendered sequence with a KNOWN camera trajectory and real parallax: two textured fronto-parallel planes (the near
    one covers the central rectangle of frame 0); under a camera that rolls by `roll` rad and moves by `step` per frame
    the image motion of the plane at depth Z is the similarity  p' - c = Z / (Z + Tz) R2 (p - c) + f T_xy / (Z + Tz),
    which render_frame renders exactly.  -> frames uint8 [n, h, w], K (principal point at the image centre),
    poses [n, 4, 4] mapping frame-0 camera coordinates to frame-t camera coordinates."""
    K = np.array([[f, 0, (w - 1) / 2], [0, f, (h - 1) / 2], [0, 0, 1]])
    c = K[:2, 2]
    tex_bg = make_texture(h + 2 * margin, w + 2 * margin, seed)
    tex_fg = make_texture(h + 2 * margin, w + 2 * margin, seed + 1)
    rect = (0.29 * w, 0.25 * h, 0.72 * w, 0.77 * h)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    frames = np.empty((n_frames, h, w), np.uint8)
    poses = np.empty((n_frames, 4, 4))
    for t in range(n_frames):
        ang = roll * t
        Hm = np.eye(4)
        Hm[:2, :2] = [[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]
        Hm[:3, 3] = np.asarray(step, float) * t
        poses[t] = Hm
        A = []
        for Z in (z_bg, z_fg):
            s_ = Z / (Z + Hm[2, 3])
            Ak = np.zeros((2, 3)); Ak[:, :2] = s_ * Hm[:2, :2]; Ak[:, 2] = c - s_ * Hm[:2, :2] @ c + f * Hm[:2, 3] / (Z + Hm[2, 3])
            A.append(Ak)
        bg = render_frame(tex_bg, A[0], w, h, margin)
        fg = render_frame(tex_fg, A[1], w, h, margin)
        Ainv = np.linalg.inv(np.vstack([A[1], [0, 0, 1]]))[:2]
        x0 = Ainv[0, 0] * xs + Ainv[0, 1] * ys + Ainv[0, 2]; y0 = Ainv[1, 0] * xs + Ainv[1, 1] * ys + Ainv[1, 2]
        inside = (x0 >= rect[0]) & (x0 < rect[2]) & (y0 >= rect[1]) & (y0 < rect[3])
        frames[t] = np.where(inside, fg, bg)
    return frames, K, poses


# ----------------------------------------------------------------------------
# closed-loop scene: two textured planes, a camera that sways back and forth (periodic: a sequence of any length stays in
# view and frame n_period follows frame n_period - 1 seamlessly), ground-truth bootstrap state
# ----------------------------------------------------------------------------
def sway_pose(t, amp=(0.9, 0.25, -0.5), roll=0.02, period=40.0):
    """frame-0 camera -> frame-t camera: the camera side-steps, rises and approaches on a sinusoid and rolls with it"""
    s = np.sin(2 * np.pi * t / period)
    Hm = np.eye(4)
    ang = roll * s
    Hm[:2, :2] = [[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]
    Hm[:3, 3] = np.asarray(amp, float) * s
    return Hm


def sway_scene(n_frames, w=416, h=240, f=400.0, seed=2024, z_bg=10.0, z_fg=6.5, pose_fn=sway_pose, margin=128):
    """-> dict(frames [n, h, w] u8, K, poses [n, 4, 4], depth(t, xy) -> Z of the surface seen at pixel xy of frame t and its frame-0 pixel)"""
    K = np.array([[f, 0, (w - 1) / 2], [0, f, (h - 1) / 2], [0, 0, 1]])
    c = K[:2, 2]
    tex_bg = make_texture(h + 2 * margin, w + 2 * margin, seed)
    tex_fg = make_texture(h + 2 * margin, w + 2 * margin, seed + 1)
    rect = (0.29 * w, 0.25 * h, 0.72 * w, 0.77 * h)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    frames = np.empty((n_frames, h, w), np.uint8)
    poses = np.empty((n_frames, 4, 4))
    motions = []

    def plane_motion(Hm, Z):
        s_ = Z / (Z + Hm[2, 3])
        A = np.zeros((2, 3)); A[:, :2] = s_ * Hm[:2, :2]; A[:, 2] = c - s_ * Hm[:2, :2] @ c + f * Hm[:2, 3] / (Z + Hm[2, 3])
        return A
    for t in range(n_frames):
        Hm = pose_fn(t)
        poses[t] = Hm
        A_bg, A_fg = plane_motion(Hm, z_bg), plane_motion(Hm, z_fg)
        motions.append((A_bg, A_fg))
        bg = render_frame(tex_bg, A_bg, w, h, margin)
        fg = render_frame(tex_fg, A_fg, w, h, margin)
        Ainv = np.linalg.inv(np.vstack([A_fg, [0, 0, 1]]))[:2]
        x0 = Ainv[0, 0] * xs + Ainv[0, 1] * ys + Ainv[0, 2]; y0 = Ainv[1, 0] * xs + Ainv[1, 1] * ys + Ainv[1, 2]
        inside = (x0 >= rect[0]) & (x0 < rect[2]) & (y0 >= rect[1]) & (y0 < rect[3])
        frames[t] = np.where(inside, fg, bg)

    def surface(t, xy):
        """for pixels xy [n, 2] of frame t: (Z [n], frame-0 pixel [n, 2]) of the plane seen there"""
        xy = np.asarray(xy, np.float64).reshape(-1, 2)
        out_z, out_p0 = np.empty(len(xy)), np.empty((len(xy), 2))
        for k, (A, Z) in enumerate(((motions[t][1], z_fg), (motions[t][0], z_bg))):
            Ainv = np.linalg.inv(np.vstack([A, [0, 0, 1]]))[:2]
            p0 = xy @ Ainv[:, :2].T + Ainv[:, 2]
            if k == 0:
                fgm = (p0[:, 0] >= rect[0]) & (p0[:, 0] < rect[2]) & (p0[:, 1] >= rect[1]) & (p0[:, 1] < rect[3])
                out_z[fgm], out_p0[fgm] = Z, p0[fgm]
            else:
                out_z[~fgm], out_p0[~fgm] = Z, p0[~fgm]
        return out_z, out_p0
    return dict(frames=frames, K=K, poses=poses, surface=surface, f=f)


def gt_bootstrap(ctx, sc, t0=0, t1=4, n_landmarks=0.6, min_kp_dist=7):
    """A State like Pipeline._get_init_state's (pipeline.py:42-90) from ground truth instead of SIFT + five-point: Shi-Tomasi corners
    of frame t1 (through `ctx`), the first `n_landmarks` share of them become landmarks at their true position (world = camera t0,
    unit = the t0 -> t1 baseline, as the bootstrap fixes it), the rest candidates born at step 1.  -> (state, t_loader)"""
    from .extractor import Extractor
    from .state import Landmark, State, Trajectory
    ext = Extractor(min_kp_dist=min_kp_dist, ctx=ctx)
    kps = ext.extract(sc["frames"][t1], 1, current_kp=[], detector='shi-tomasi', mask_radius=min_kp_dist, describe=False)
    G0, G1 = sc["poses"][t0], sc["poses"][t1]
    rel = G1 @ np.linalg.inv(G0)
    unit = np.linalg.norm(rel[:3, 3])
    H1 = rel.copy(); H1[:3, 3] /= unit
    n_l = int(len(kps) * n_landmarks) if n_landmarks <= 1 else int(n_landmarks)
    uv = np.array([k.uv.reshape(2) for k in kps[:n_l]], np.float64)
    Z, p0 = sc["surface"](t1, uv)
    K = sc["K"]
    X0 = np.stack([(p0[:, 0] - K[0, 2]) / K[0, 0] * Z, (p0[:, 1] - K[1, 2]) / K[1, 1] * Z, Z], 1)      # frame-0 camera coordinates
    Xw = (X0 @ G0[:3, :3].T + G0[:3, 3]) / unit                                                        # camera t0 = world, unit baseline
    lms = [Landmark(1, Xw[i].reshape(3, 1).copy(), kps[i].des) for i in range(n_l)]
    traj = Trajectory({})
    traj.append(0, np.eye(4)); traj.append(1, H1)
    return State(lms, kps[:n_l], kps[n_l:], traj), t1


