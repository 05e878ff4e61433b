"""sift_oracle.py -- CPU ORACLE (test infrastructure, not product code) for the bootstrap's SIFT detector / descriptor.

Reference call site: Extractor.extract(img, t, detector='custom', describe=True),
/root/reference/src/extractor/extractor.py:26-28 (cv2.SIFT_create(nfeatures=1000)), :114-122 (detect, compute);
used once per sequence by Pipeline._get_init_state, pipeline.py:48-49 (SURVEY.md 8f "next" row 4).

PARITY STATUS: *** unpinned ***.  The arithmetic lives in OpenCV 4.4.0 (modules/features2d/src/sift.dispatch.cpp,
sift.simd.hpp, imgproc smooth / resize), which is absent here, and the reference holds no vectors for it.  This file
restates that implementation from its published structure (Lowe 2004 + OpenCV's constants and control flow):
  createInitialImage (float image, 2x INTER_LINEAR upsample, blur to sigma 1.6), buildGaussianPyramid (nOctaveLayers + 3
  images per octave, incremental separable Gaussian blurs with kernel size cvRound(8 sigma + 1) | 1, REFLECT_101, next
  octave by INTER_NEAREST decimation), DoG, 26-neighbour extrema above floor(0.5 * 0.04 / 3 * 255), adjustLocalExtrema
  (<= 5 quadratic-fit steps, contrast and edge tests), calcOrientationHist (36 bins, fastAtan2 polynomial, [1 4 6 4 1] / 16
  smoothing, peaks >= 0.8 max with parabolic interpolation), KeyPointsFilter::removeDuplicatedSorted + retainBest(1000),
  first-octave rescale, calcSIFTDescriptor (4 x 4 x 8 trilinear histogram, 0.2 clamp, x 512, saturate to uchar).
Deviations that cannot be resolved without OpenCV itself (rounding level only): float summation order inside the separable
filter (here: centre tap, then symmetric pairs outward), the Gaussian kernel from double exp instead of softdouble,
exp / cos / sin / pow evaluated in float64 and rounded to float32 (OpenCV: its own float32 SIMD approximations), the
3 x 3 solve by Cramer's rule in float32.  fastAtan2 is OpenCV's own polynomial.  The final keypoint ORDER (unspecified in
OpenCV after nth_element) is the removeDuplicatedSorted order.
This file DEFINES what csrc/vo_sift.hip implements; every float32 operation is written so that the GPU can follow it
operation by operation (no fused multiply-add, sequential histogram accumulation in sample order).
"""
import math

import numpy as np

f32 = np.float32

N_LAYERS = 3
SIGMA = 1.6
CONTRAST_THR = 0.04
EDGE_THR = 10.0
IMG_BORDER = 5
MAX_INTERP_STEPS = 5
ORI_BINS = 36
ORI_SIG_FCTR = f32(1.5)
ORI_RADIUS = f32(3.0) * ORI_SIG_FCTR
ORI_PEAK_RATIO = f32(0.8)
DESCR_WIDTH = 4
DESCR_BINS = 8
DESCR_SCL_FCTR = f32(3.0)
DESCR_MAG_THR = f32(0.2)
INT_DESCR_FCTR = f32(512.0)
FLT_EPSILON = f32(1.1920929e-07)


def cv_round(x):
    return int(np.rint(x))


def gaussian_kernel(sigma):
    """cv::getGaussianKernel(ksize from sigma for CV_32F, sigma) -> float32 taps"""
    n = cv_round(sigma * 4 * 2 + 1) | 1
    x = np.arange(n, dtype=np.float64) - (n - 1) * 0.5
    t = np.exp(-0.5 / (sigma * sigma) * x * x)
    return (t / t.sum()).astype(f32)


def blur(img, sigma):
    """separable Gaussian blur, float32, BORDER_REFLECT_101: rows then columns; per pixel centre tap first, then the
    symmetric pairs outward: s = k0 x0; s += k_i (x_-i + x_+i)"""
    k = gaussian_kernel(sigma)
    r = len(k) // 2
    out = img
    for axis in (1, 0):
        n = out.shape[axis]
        idx = np.arange(n)
        src = out
        acc = k[r] * src
        for i in range(1, r + 1):
            lo = np.take(src, _reflect_idx(idx - i, n), axis=axis)
            hi = np.take(src, _reflect_idx(idx + i, n), axis=axis)
            acc = acc + k[r + i] * (lo + hi)
        out = acc
    return out


def _reflect_idx(i, n):
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)                       # kernel radii can exceed tiny top-octave images: full periodic reflection
    return np.where(i >= n, p - i, i)


def upsample2(img):
    """cv::resize(img, (2w, 2h), INTER_LINEAR) for float32: sample position (x + 0.5) / 2 - 0.5, clamped"""
    h, w = img.shape

    def taps(n):
        d = np.arange(2 * n)
        fx = (d.astype(np.float64) + 0.5) * 0.5 - 0.5
        s = np.floor(fx).astype(np.int64)
        a = (fx - s).astype(f32)
        a = np.where(s < 0, f32(0), a); s0 = np.maximum(s, 0)
        a = np.where(s0 >= n - 1, f32(0), a); s0 = np.minimum(s0, n - 1)
        s1 = np.minimum(s0 + 1, n - 1)
        return s0, s1, (f32(1) - a).astype(f32), a.astype(f32)
    x0, x1, wx0, wx1 = taps(w)
    y0, y1, wy0, wy1 = taps(h)
    rows = img[:, x0] * wx0[None, :] + img[:, x1] * wx1[None, :]            # horizontal pass, float32
    return rows[y0, :] * wy0[:, None] + rows[y1, :] * wy1[:, None]


def decimate(img):
    """cv::resize(img, (w / 2, h / 2), INTER_NEAREST): source index min(floor(x * (1 / (dst / src))), src - 1)"""
    h, w = img.shape
    dh, dw = h // 2, w // 2
    ifx, ify = 1.0 / (dw / w), 1.0 / (dh / h)
    xs = np.minimum(np.floor(np.arange(dw) * ifx).astype(np.int64), w - 1)
    ys = np.minimum(np.floor(np.arange(dh) * ify).astype(np.int64), h - 1)
    return img[np.ix_(ys, xs)]


def layer_sigmas():
    k = 2.0 ** (1.0 / N_LAYERS)
    sig = [SIGMA]
    for i in range(1, N_LAYERS + 3):
        sp = k ** (i - 1) * SIGMA
        st = sp * k
        sig.append(math.sqrt(st * st - sp * sp))
    return sig


def n_octaves(w, h):
    return cv_round(math.log(min(2 * w, 2 * h)) / math.log(2.0) - 2) + 1


def build_pyramids(img):
    """-> gauss[o][0..5], dog[o][0..4] (float32), octave 0 = the doubled image"""
    g = img.astype(f32)
    sig_diff = math.sqrt(max(SIGMA * SIGMA - 0.5 * 0.5 * 4, 0.01))
    base = blur(upsample2(g), sig_diff)
    sig = layer_sigmas()
    gauss, dog = [], []
    for o in range(n_octaves(img.shape[1], img.shape[0])):
        if o > 0:
            base = decimate(gauss[o - 1][N_LAYERS])
        if min(base.shape) < 1:
            break
        layers = [base]
        for i in range(1, N_LAYERS + 3):
            layers.append(blur(layers[-1], sig[i]))
        gauss.append(layers)
        dog.append([layers[i + 1] - layers[i] for i in range(N_LAYERS + 2)])
    return gauss, dog


def fast_atan2_vec(y, x):
    """cv::fastAtan2 (degrees): OpenCV's own float32 polynomial, constants = float coefficient x (float)(180 / pi)"""
    deg = f32(180 / math.pi)
    p1 = f32(f32(0.9997878412794807) * deg); p3 = f32(f32(-0.3258083974640975) * deg)
    p5 = f32(f32(0.1555786518463281) * deg); p7 = f32(f32(-0.04432655554792128) * deg)
    eps = f32(2.220446049250313e-16)
    ax, ay = np.abs(x), np.abs(y)
    big = ax >= ay
    num = np.where(big, ay, ax); den = np.where(big, ax, ay) + eps
    c = num / den; c2 = c * c
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    a = np.where(big, a, f32(90.0) - a)
    a = np.where(x < 0, f32(180.0) - a, a)
    a = np.where(y < 0, f32(360.0) - a, a)
    return a.astype(f32)


def exp_f32(w):
    """float32 exp through float64 (order-independent, reproducible on the GPU)"""
    return np.exp(np.asarray(w, np.float64)).astype(f32)


def solve3(H, b):
    """Cramer's rule in float32; None if singular"""
    a = H
    d = f32(f32(a[0][0] * f32(f32(a[1][1] * a[2][2]) - f32(a[1][2] * a[2][1]))) - f32(a[0][1] * f32(f32(a[1][0] * a[2][2]) - f32(a[1][2] * a[2][0])))
            + f32(a[0][2] * f32(f32(a[1][0] * a[2][1]) - f32(a[1][1] * a[2][0]))))
    if d == 0:
        return None
    d = f32(f32(1.0) / d)
    x0 = f32(d * f32(f32(f32(b[0] * f32(f32(a[1][1] * a[2][2]) - f32(a[1][2] * a[2][1]))) - f32(a[0][1] * f32(f32(b[1] * a[2][2]) - f32(a[1][2] * b[2]))))
                     + f32(a[0][2] * f32(f32(b[1] * a[2][1]) - f32(a[1][1] * b[2])))))
    x1 = f32(d * f32(f32(f32(a[0][0] * f32(f32(b[1] * a[2][2]) - f32(a[1][2] * b[2]))) - f32(b[0] * f32(f32(a[1][0] * a[2][2]) - f32(a[1][2] * a[2][0]))))
                     + f32(a[0][2] * f32(f32(a[1][0] * b[2]) - f32(b[1] * a[2][0])))))
    x2 = f32(d * f32(f32(f32(a[0][0] * f32(f32(a[1][1] * b[2]) - f32(b[1] * a[2][1]))) - f32(a[0][1] * f32(f32(a[1][0] * b[2]) - f32(b[1] * a[2][0]))))
                     + f32(b[0] * f32(f32(a[1][0] * a[2][1]) - f32(a[1][1] * a[2][0])))))
    return x0, x1, x2


def adjust_local_extrema(dog_o, octv, layer, r, c):
    """-> None or dict(x, y, octave, size, response, layer, r, c) in octave-0 (doubled image) coordinates"""
    img_scale = f32(1.0 / 255.0)
    deriv_scale = f32(img_scale * f32(0.5)); second = img_scale; cross = f32(img_scale * f32(0.25))
    xi = xr = xc = f32(0)
    i = 0
    while i < MAX_INTERP_STEPS:
        img, prev, nxt = dog_o[layer], dog_o[layer - 1], dog_o[layer + 1]
        dD = (f32(f32(img[r, c + 1] - img[r, c - 1]) * deriv_scale), f32(f32(img[r + 1, c] - img[r - 1, c]) * deriv_scale),
              f32(f32(nxt[r, c] - prev[r, c]) * deriv_scale))
        v2 = f32(img[r, c] * f32(2))
        dxx = f32(f32(f32(img[r, c + 1] + img[r, c - 1]) - v2) * second)
        dyy = f32(f32(f32(img[r + 1, c] + img[r - 1, c]) - v2) * second)
        dss = f32(f32(f32(nxt[r, c] + prev[r, c]) - v2) * second)
        dxy = f32(f32(f32(f32(img[r + 1, c + 1] - img[r + 1, c - 1]) - img[r - 1, c + 1]) + img[r - 1, c - 1]) * cross)
        dxs = f32(f32(f32(f32(nxt[r, c + 1] - nxt[r, c - 1]) - prev[r, c + 1]) + prev[r, c - 1]) * cross)
        dys = f32(f32(f32(f32(nxt[r + 1, c] - nxt[r - 1, c]) - prev[r + 1, c]) + prev[r - 1, c]) * cross)
        X = solve3(((dxx, dxy, dxs), (dxy, dyy, dys), (dxs, dys, dss)), dD)
        if X is None:
            X = (f32(0), f32(0), f32(0))                    # Matx::solve leaves the zero vector when it fails
        xi, xr, xc = f32(-X[2]), f32(-X[1]), f32(-X[0])
        if abs(xi) < 0.5 and abs(xr) < 0.5 and abs(xc) < 0.5:
            break
        if abs(xi) > 2147483647 / 3 or abs(xr) > 2147483647 / 3 or abs(xc) > 2147483647 / 3:
            return None
        c += cv_round(xc); r += cv_round(xr); layer += cv_round(xi)
        rows, cols = img.shape
        if layer < 1 or layer > N_LAYERS or c < IMG_BORDER or c >= cols - IMG_BORDER or r < IMG_BORDER or r >= rows - IMG_BORDER:
            return None
        i += 1
    if i >= MAX_INTERP_STEPS:
        return None
    img, prev, nxt = dog_o[layer], dog_o[layer - 1], dog_o[layer + 1]
    dD = (f32(f32(img[r, c + 1] - img[r, c - 1]) * deriv_scale), f32(f32(img[r + 1, c] - img[r - 1, c]) * deriv_scale),
          f32(f32(nxt[r, c] - prev[r, c]) * deriv_scale))
    t = f32(f32(f32(dD[0] * xc) + f32(dD[1] * xr)) + f32(dD[2] * xi))
    contr = f32(f32(img[r, c] * img_scale) + f32(t * f32(0.5)))
    if float(f32(abs(contr) * f32(N_LAYERS))) < CONTRAST_THR:          # float product against the double threshold
        return None
    v2 = f32(img[r, c] * f32(2))
    dxx = f32(f32(f32(img[r, c + 1] + img[r, c - 1]) - v2) * second)
    dyy = f32(f32(f32(img[r + 1, c] + img[r - 1, c]) - v2) * second)
    dxy = f32(f32(f32(f32(img[r + 1, c + 1] - img[r + 1, c - 1]) - img[r - 1, c + 1]) + img[r - 1, c - 1]) * cross)
    tr = f32(dxx + dyy)
    det = f32(f32(dxx * dyy) - f32(dxy * dxy))
    if det <= 0 or float(f32(tr * tr)) * EDGE_THR >= (EDGE_THR + 1) * (EDGE_THR + 1) * float(det):   # in double, as in OpenCV
        return None
    sc = f32(1 << octv)
    pw = f32(2.0 ** float(f32(f32(f32(layer) + xi) / f32(N_LAYERS))))           # powf
    size = f32(SIGMA * float(pw) * (1 << octv) * 2)                              # double product (sigma is a double), stored as float
    return dict(x=f32(f32(f32(c) + xc) * sc), y=f32(f32(f32(r) + xr) * sc),
                octave=octv + (layer << 8) + (cv_round((float(xi) + 0.5) * 255) << 16),
                size=size, response=f32(abs(contr)), layer=layer, r=r, c=c)


def orientation_hist(img, px, py, radius, sigma):
    """calcOrientationHist -> smoothed 36-bin histogram (float32) and its maximum"""
    n = ORI_BINS
    rows, cols = img.shape
    expf_scale = f32(f32(-1.0) / f32(f32(2.0) * f32(sigma * sigma)))
    ii, jj = np.meshgrid(np.arange(-radius, radius + 1), np.arange(-radius, radius + 1), indexing="ij")
    ii = ii.ravel(); jj = jj.ravel()                       # sample order: rows, then columns
    y = py + ii; x = px + jj
    ok = (y > 0) & (y < rows - 1) & (x > 0) & (x < cols - 1)
    ii, jj, y, x = ii[ok], jj[ok], y[ok], x[ok]
    dx = img[y, x + 1] - img[y, x - 1]
    dy = img[y - 1, x] - img[y + 1, x]
    W = exp_f32((ii * ii + jj * jj).astype(f32) * expf_scale)
    ori = fast_atan2_vec(dy, dx)
    mag = np.sqrt(dx * dx + dy * dy)
    b = np.rint(f32(n / 360.0) * ori).astype(np.int64)
    b = np.where(b >= n, b - n, b); b = np.where(b < 0, b + n, b)
    temphist = np.zeros(n, f32)
    np.add.at(temphist, b, W * mag)                         # sequential, float32, in sample order
    t = np.concatenate([temphist[-2:], temphist, temphist[:2]])
    hist = ((t[0:n] + t[4:n + 4]) * f32(1.0 / 16.0) + (t[1:n + 1] + t[3:n + 3]) * f32(4.0 / 16.0)) + t[2:n + 2] * f32(6.0 / 16.0)
    return hist.astype(f32), hist.max()


def find_extrema(gauss, dog):
    """findScaleSpaceExtrema -> list of keypoint dicts (doubled-image coordinates), in (octave, layer, row, column) order"""
    thr = math.floor(0.5 * CONTRAST_THR / N_LAYERS * 255)
    kps = []
    for o in range(len(dog)):
        rows, cols = dog[o][0].shape
        if rows <= 2 * IMG_BORDER or cols <= 2 * IMG_BORDER:
            continue
        stack = np.stack(dog[o])                             # [5][rows][cols]
        for layer in range(1, N_LAYERS + 1):
            v = stack[layer, IMG_BORDER:rows - IMG_BORDER, IMG_BORDER:cols - IMG_BORDER]
            mx = np.full(v.shape, -np.inf, f32); mn = np.full(v.shape, np.inf, f32)
            for dl in (-1, 0, 1):
                for dr in (-1, 0, 1):
                    for dc in (-1, 0, 1):
                        nb = stack[layer + dl, IMG_BORDER + dr:rows - IMG_BORDER + dr, IMG_BORDER + dc:cols - IMG_BORDER + dc]
                        mx = np.maximum(mx, nb); mn = np.minimum(mn, nb)
            cand = (np.abs(v) > thr) & (((v > 0) & (v >= mx)) | ((v < 0) & (v <= mn)))
            for r, c in zip(*np.nonzero(cand)):
                k = adjust_local_extrema(dog[o], o, layer, int(r) + IMG_BORDER, int(c) + IMG_BORDER)
                if k is None:
                    continue
                scl_octv = f32(f32(k["size"] * f32(0.5)) / f32(1 << o))
                hist, omax = orientation_hist(gauss[o][k["layer"]], k["c"], k["r"], cv_round(f32(ORI_RADIUS * scl_octv)),
                                              f32(ORI_SIG_FCTR * scl_octv))
                mag_thr = f32(omax * ORI_PEAK_RATIO)
                n = ORI_BINS
                for j in range(n):
                    l = j - 1 if j > 0 else n - 1
                    r2 = j + 1 if j < n - 1 else 0
                    if hist[j] > hist[l] and hist[j] > hist[r2] and hist[j] >= mag_thr:
                        b = f32(f32(j) + f32(f32(f32(0.5) * f32(hist[l] - hist[r2])) / f32(f32(hist[l] - f32(f32(2) * hist[j])) + hist[r2])))
                        b = f32(n + b) if b < 0 else (f32(b - n) if b >= n else b)
                        ang = f32(f32(360.0) - f32(f32(360.0 / n) * b))
                        if abs(f32(ang - f32(360.0))) < FLT_EPSILON:
                            ang = f32(0)
                        kps.append(dict(k, angle=ang))
    return kps


def filter_keypoints(kps, nfeatures=1000):
    """KeyPointsFilter::removeDuplicatedSorted + retainBest + the first-octave (-1) rescale"""
    key = lambda k: (float(k["x"]), float(k["y"]), -float(k["size"]), float(k["angle"]), -float(k["response"]), -k["octave"])
    kps = sorted(kps, key=key)
    out = []
    for k in kps:
        if out and (out[-1]["x"], out[-1]["y"], out[-1]["size"], out[-1]["angle"]) == (k["x"], k["y"], k["size"], k["angle"]):
            continue
        out.append(k)
    if nfeatures > 0 and len(out) > nfeatures:
        thr = sorted((float(k["response"]) for k in out), reverse=True)[nfeatures - 1]
        out = [k for k in out if float(k["response"]) >= thr]
    res = []
    for k in out:
        octave = (k["octave"] & ~255) | ((k["octave"] - 1) & 255)
        res.append(dict(x=f32(k["x"] * f32(0.5)), y=f32(k["y"] * f32(0.5)), size=f32(k["size"] * f32(0.5)), angle=k["angle"],
                        response=k["response"], octave=octave))
    return res


def unpack_octave(octave):
    o = octave & 255
    layer = (octave >> 8) & 255
    o = o if o < 128 else (-128 | o)
    scale = f32(1.0 / (1 << o)) if o >= 0 else f32(1 << -o)
    return o, layer, scale


def descriptor(img, ptx, pty, ori, scl):
    """calcSIFTDescriptor -> 128 float32 values in 0..255"""
    d, n = DESCR_WIDTH, DESCR_BINS
    rows, cols = img.shape
    px, py = cv_round(ptx), cv_round(pty)
    ang = float(f32(ori * f32(math.pi / 180.0)))
    cos_t = f32(math.cos(ang)); sin_t = f32(math.sin(ang))
    bins_per_rad = f32(n / 360.0)
    exp_scale = f32(f32(-1.0) / f32(d * d * 0.5))
    hist_width = f32(DESCR_SCL_FCTR * scl)
    radius = cv_round(f32(f32(f32(hist_width * f32(1.4142135623730951)) * f32(d + 1)) * f32(0.5)))
    radius = min(radius, int(math.sqrt(float(cols) * cols + float(rows) * rows)))
    cos_t = f32(cos_t / hist_width); sin_t = f32(sin_t / hist_width)
    ii, jj = np.meshgrid(np.arange(-radius, radius + 1), np.arange(-radius, radius + 1), indexing="ij")
    ii = ii.ravel(); jj = jj.ravel()
    fi, fj = ii.astype(f32), jj.astype(f32)
    c_rot = fj * cos_t - fi * sin_t
    r_rot = fj * sin_t + fi * cos_t
    rbin = (r_rot + f32(d // 2)) - f32(0.5)
    cbin = (c_rot + f32(d // 2)) - f32(0.5)
    r = py + ii; c = px + jj
    ok = (rbin > -1) & (rbin < d) & (cbin > -1) & (cbin < d) & (r > 0) & (r < rows - 1) & (c > 0) & (c < cols - 1)
    rbin, cbin, r, c, c_rot, r_rot = rbin[ok], cbin[ok], r[ok], c[ok], c_rot[ok], r_rot[ok]
    dx = img[r, c + 1] - img[r, c - 1]
    dy = img[r - 1, c] - img[r + 1, c]
    W = exp_f32((c_rot * c_rot + r_rot * r_rot) * exp_scale)
    Ori = fast_atan2_vec(dy, dx)
    Mag = np.sqrt(dx * dx + dy * dy)
    obin = (Ori - f32(ori)) * bins_per_rad
    mag = Mag * W
    r0 = np.floor(rbin).astype(np.int64); c0 = np.floor(cbin).astype(np.int64); o0 = np.floor(obin).astype(np.int64)
    rbin = rbin - r0.astype(f32); cbin = cbin - c0.astype(f32); obin = obin - o0.astype(f32)
    o0 = np.where(o0 < 0, o0 + n, o0); o0 = np.where(o0 >= n, o0 - n, o0)
    v_r1 = mag * rbin; v_r0 = mag - v_r1
    v_rc11 = v_r1 * cbin; v_rc10 = v_r1 - v_rc11
    v_rc01 = v_r0 * cbin; v_rc00 = v_r0 - v_rc01
    v111 = v_rc11 * obin; v110 = v_rc11 - v111
    v101 = v_rc10 * obin; v100 = v_rc10 - v101
    v011 = v_rc01 * obin; v010 = v_rc01 - v011
    v001 = v_rc00 * obin; v000 = v_rc00 - v001
    idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0
    offs = [0, 1, n + 2, n + 3, (d + 2) * (n + 2), (d + 2) * (n + 2) + 1, (d + 3) * (n + 2), (d + 3) * (n + 2) + 1]
    vals = [v000, v001, v010, v011, v100, v101, v110, v111]
    all_idx = np.stack([idx + o for o in offs], 1).ravel()          # per sample its 8 bins, samples in order
    all_val = np.stack(vals, 1).ravel().astype(f32)
    hist = np.zeros((d + 2) * (d + 2) * (n + 2), f32)
    np.add.at(hist, all_idx, all_val)
    dst = np.zeros(d * d * n, f32)
    for i in range(d):
        for j in range(d):
            b = ((i + 1) * (d + 2) + (j + 1)) * (n + 2)
            hist[b] = hist[b] + hist[b + n]
            hist[b + 1] = hist[b + 1] + hist[b + n + 1]
            dst[(i * d + j) * n:(i * d + j + 1) * n] = hist[b:b + n]
    nrm2 = f32(0)
    for v in dst:
        nrm2 = f32(nrm2 + f32(v * v))
    thr = f32(f32(np.sqrt(nrm2)) * DESCR_MAG_THR)
    nrm2 = f32(0)
    for k in range(len(dst)):
        val = min(dst[k], thr)
        dst[k] = val
        nrm2 = f32(nrm2 + f32(val * val))
    scale = f32(INT_DESCR_FCTR / max(f32(np.sqrt(nrm2)), FLT_EPSILON))
    return np.clip(np.rint(dst * scale), 0, 255).astype(f32)


def detect_and_compute(img, nfeatures=1000, mask=None):
    """cv2.SIFT_create(nfeatures).detect(img, mask) + .compute(img, kps)
    -> keypoints (n, 6) float64 [x, y, size, angle, response, octave], descriptors (n, 128) float32"""
    img = np.asarray(img, np.uint8)
    gauss, dog = build_pyramids(img)
    kps = filter_keypoints(find_extrema(gauss, dog), nfeatures)
    if mask is not None:
        kps = [k for k in kps if mask[int(k["y"] + f32(0.5)), int(k["x"] + f32(0.5))] != 0]
    desc = np.zeros((len(kps), 128), f32)
    for i, k in enumerate(kps):
        o, layer, scale = unpack_octave(k["octave"])
        size = f32(k["size"] * scale)
        ang = f32(f32(360.0) - k["angle"])
        if abs(f32(ang - f32(360.0))) < FLT_EPSILON:
            ang = f32(0)
        desc[i] = descriptor(gauss[o + 1][layer], f32(k["x"] * scale), f32(k["y"] * scale), ang, f32(size * f32(0.5)))
    arr = np.array([[k["x"], k["y"], k["size"], k["angle"], k["response"], k["octave"]] for k in kps], np.float64).reshape(-1, 6)
    return arr, desc
