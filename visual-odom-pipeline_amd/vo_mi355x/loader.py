"""Dataset access behind the reference's `Loader` interface (src/loader/loader.py:11-108): `Loader(name, cfg)` with
getImage / getPose / getFrame / getCamera / getInit, len() and str().

Every dataset the reference knows is ONE ROW of `LAYOUTS` -- where the frames live, how the camera matrix is stored and
whether ground-truth poses exist -- read by one generic routine.  Frames are decoded with PIL (`imread_gray` documents how
that relates to cv2.imread(IMREAD_GRAYSCALE)); the pre-filter cv2.bilateralFilter(d=5, sigmaColor=1.5, sigmaSpace=1.5) that
the reference applies to every frame (:16-20, :86) runs on the GPU (`VoContext.bilateral`, the kernel fused into the
frame store's level-0 pass; bit-identical to the oracle's restatement, tests/test_gpu_prefilter.py).
"""
from collections import namedtuple
from pathlib import Path

import numpy as np

PREFILTER = dict(d=5, sigmaColor=1.5, sigmaSpace=1.5)                    # reference loader.py:16-20


def _number_rows(path):
    """rows of numbers separated by commas and / or blanks (K.txt of the VAMR sets ends its rows with a comma)"""
    with open(path) as f:
        rows = [[float(v) for v in line.replace(",", " ").split()] for line in f]
    return np.array([r for r in rows if r])


def _k_file(base):
    return _number_rows(base / "K.txt")[:3, :3]


def _k_kitti(base):
    """first row of calib.txt: 'P0:' followed by the 3 x 4 projection matrix of camera 0"""
    with open(base / "00" / "calib.txt") as f:
        vals = [float(v) for v in f.readline().split()[1:13]]
    return np.array(vals).reshape(3, 4)[:, :3]


def _k_malaga(base):
    """'key=value' lines of the rectified-camera file; the left camera's cx, cy, fx, fy are lines 7-10 of the file"""
    with open(base / "camera_params_rectified_a=0_1024x768.txt") as f:
        lines = f.readlines()
    val = {name: float(lines[row].split("=", 1)[1]) for name, row in (("cx", 6), ("cy", 7), ("fx", 8), ("fy", 9))}
    return np.array([[val["fx"], 0.0, val["cx"]], [0.0, val["fy"], val["cy"]], [0.0, 0.0, 1.0]])


# frames: (sub-directory, glob); camera: reader(base) -> 3 x 3; poses: file of 3 x 4 row-major matrices, or None;
# unit_w: whether pose-less sets get H[3, 3] = 1 (the reference leaves malaga's matrices all zero, loader.py:56)
Layout = namedtuple("Layout", "frames camera poses unit_w")
_FOLDER = Layout(("images", "*.png"), _k_file, None, True)
LAYOUTS = {
    "parking": Layout(("images", "*.png"), _k_file, "poses.txt", True),
    "kitti": Layout(("00/image_0", "*.png"), _k_kitti, "poses/00.txt", True),
    "malaga": Layout(("malaga-urban-dataset-extract-07_rectified_1024x768_Images", "*_left.jpg"), _k_malaga, None, False),
    "roomtour": _FOLDER, "stairway": _FOLDER, "outdoor_street": _FOLDER, "outdoor_loop": _FOLDER,
}


def imread_gray(path):
    """cv2.imread(path, cv2.IMREAD_GRAYSCALE) with PIL.
    * 8-bit grey PNG: the stored samples (identical).
    * colour PNG: OpenCV's fixed-point BGR2GRAY, (4899 R + 9617 G + 1868 B + 8192) >> 14 (identical).
    * 16-bit grey PNG: OpenCV scales to 8 bit by >> 8 (PIL's convert('L') would clip instead).
    * JPEG: OpenCV lets libjpeg emit the luma plane itself (JCS_GRAYSCALE); PIL's draft('L') asks libjpeg for the same
      thing.  Both sit on libjpeg's IDCT, so the planes are equal when the two link the same libjpeg build -- parity for
      the 'malaga' set is therefore decoder-dependent and not pinned here."""
    from PIL import Image
    im = Image.open(path)
    if im.format == "JPEG":
        im.draft("L", im.size)
        return np.ascontiguousarray(np.asarray(im.convert("L")), np.uint8)
    if im.mode in ("I;16", "I;16B", "I"):
        return np.ascontiguousarray(np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8)
    if im.mode in ("L", "P", "1"):
        return np.ascontiguousarray(np.asarray(im.convert("L")), np.uint8)
    rgb = np.asarray(im.convert("RGB")).astype(np.int32)
    return ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


class Loader:
    def __init__(self, name, cfg, ctx=None, device=0):
        if name not in LAYOUTS:
            raise Exception("unknown dataset %r" % (name,))            # the reference raises a bare Exception (loader.py:78)
        lay, base = LAYOUTS[name], Path(cfg[name]["path"])
        self._name, self._cfg = name, cfg
        self._ctx, self._device = ctx, device
        self.image_paths = sorted(str(p) for p in (base / lay.frames[0]).rglob(lay.frames[1]))
        self._camera = lay.camera(base)
        if lay.poses is not None:
            top = np.loadtxt(str(base / lay.poses)).reshape(-1, 3, 4)
            self._poses = np.concatenate([top, np.broadcast_to([[[0.0, 0.0, 0.0, 1.0]]], (len(top), 1, 4))], axis=1)
        else:
            self._poses = np.zeros((len(self.image_paths), 4, 4))
            if lay.unit_w:
                self._poses[:, 3, 3] = 1.0

    def __str__(self):
        return self._name

    def __len__(self):
        return len(self._poses)                                         # the reference counts poses, not files (loader.py:22)

    def _at(self, seq, idx):
        if not 0 <= idx < len(self):
            raise AssertionError("frame index %r outside 0..%d" % (idx, len(self) - 1))    # reference: `assert` (:84, :90, :96)
        return seq[idx]

    def _prefilter_context(self, shape):
        if self._ctx is None:
            from .context import VoContext
            self._ctx = VoContext(shape[1], shape[0], max_pts=64, device=self._device)
        return self._ctx

    def getImage(self, id):
        raw = imread_gray(self._at(self.image_paths, id))
        return self._prefilter_context(raw.shape).bilateral(raw, PREFILTER["d"], PREFILTER["sigmaColor"], PREFILTER["sigmaSpace"])

    def getRawImage(self, id, out=None):
        """the frame as decoded, WITHOUT the pre-filter -- for a context that applies it while the frame enters the frame store
        (`VoContext.set_prefilter(**...)`: fused into the level-0 pass, bit-identical to `getImage`).  `out`: a uint8 [h, w] array to decode into, e.g. a
        page-locked one (`VoContext.host_alloc`) that `frame_step_host` / `ResidentPipeline.step_host` then read over PCIe without a staging copy."""
        raw = imread_gray(self._at(self.image_paths, id))
        if out is None:
            return raw
        if out.shape != raw.shape or out.dtype != np.uint8:
            raise ValueError("getRawImage: out must be uint8 %r" % (raw.shape,))
        out[...] = raw
        return out

    def getPose(self, id):
        return self._at(self._poses, id)

    def getFrame(self, id):
        return self.getImage(id), self.getPose(id)

    def getCamera(self):
        return self._camera

    def getInit(self):
        return tuple(self._cfg[self._name]["init"])
