"""Drop-in for the reference's `Loader` (src/loader/loader.py:11-108): same constructor (name, cfg), the same dataset
layouts ('parking', 'kitti', 'malaga', the image-folder sets), the same accessors (getImage / getPose / getFrame /
getCamera / getInit, len()).  File decoding is PIL instead of cv2.imread (8-bit grey PNGs decode identically; colour
files are converted with OpenCV's fixed-point BGR2GRAY weights); the pre-filter cv2.bilateralFilter(d=5, 1.5, 1.5)
of getImage (:16-20, :86) runs on the GPU through a context of the loader's own (bit-identical to the oracle's
restatement of OpenCV's bilateralFilter_8u, tests/test_gpu_prefilter.py).
"""
from pathlib import Path

import numpy as np


def _read_matrix(path):
    """rows of numbers separated by commas and / or blanks (K.txt of the VAMR sets has trailing commas)"""
    rows = []
    with open(path) as f:
        for line in f:
            vals = [v for v in line.replace(",", " ").split() if v]
            if vals:
                rows.append([float(v) for v in vals])
    return np.array(rows)


def imread_gray(path):
    """cv2.imread(path, cv2.IMREAD_GRAYSCALE) for 8-bit images"""
    from PIL import Image
    im = Image.open(path)
    if im.mode in ("L", "P", "1", "I;16", "I"):
        return np.ascontiguousarray(np.asarray(im.convert("L")), np.uint8)
    rgb = np.asarray(im.convert("RGB")).astype(np.int32)
    # OpenCV's 8-bit BGR2GRAY: (R 4899 + G 9617 + B 1868 + 8192) >> 14
    return ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


class Loader:
    def __init__(self, name, cfg, ctx=None, device=0):
        self._name = name
        self._cfg = cfg
        self._bilateral_filter_params = {'d': 5, 'sigmaColor': 1.5, 'sigmaSpace': 1.5}
        self._ctx, self._device = ctx, device
        self._camera, self._poses, self.image_paths = self._loadData()
        self._length = self._poses.shape[0]

    def __str__(self):
        return self._name

    def __len__(self):
        return self._length

    def _loadData(self):
        cfg = self._cfg[self._name]
        base = cfg['path']
        if self._name == 'parking':
            self._image_paths = [str(p) for p in Path(base + '/images').rglob('*.png')]
            ar = np.reshape(np.loadtxt(base + '/poses.txt'), (-1, 3, 4))
            self._poses = np.zeros((ar.shape[0], 4, 4))
            self._poses[:, 3, 3] = 1
            self._poses[:, :3, :] = ar
            self._camera = _read_matrix(base + '/K.txt')[:3, :3]
        elif self._name in ['roomtour', 'stairway', 'outdoor_street', 'outdoor_loop']:
            self._image_paths = [str(p) for p in Path(base + '/images').rglob('*.png')]
            self._poses = np.zeros((len(self._image_paths), 4, 4))
            self._poses[:, 3, 3] = 1
            self._camera = _read_matrix(base + '/K.txt')[:3, :3]
        elif self._name == 'malaga':
            p = base + '/malaga-urban-dataset-extract-07_rectified_1024x768_Images'
            self._image_paths = [str(p) for p in Path(p).rglob('*_left.jpg')]
            self._poses = np.zeros((len(self._image_paths), 4, 4))
            self._camera = np.eye(3)
            with open(base + '/camera_params_rectified_a=0_1024x768.txt') as param_f:
                lines = param_f.readlines()
                self._camera[0, 0] = float(lines[8][3:-1])
                self._camera[0, 2] = float(lines[6][3:-1])
                self._camera[1, 1] = float(lines[9][3:-1])
                self._camera[1, 2] = float(lines[7][3:-1])
        elif self._name == 'kitti':
            self._image_paths = [str(p) for p in Path(base + '/00/image_0').rglob('*.png')]
            self._camera = np.genfromtxt(base + '/00/calib.txt')[0, 1:].reshape((3, 4))[:, :3]
            ar = np.reshape(np.loadtxt(base + '/poses/00.txt'), (-1, 3, 4))
            self._poses = np.zeros((ar.shape[0], 4, 4))
            self._poses[:, 3, 3] = 1
            self._poses[:, :3, :] = ar
        else:
            raise Exception
        self._image_paths.sort()
        return self._camera, self._poses, self._image_paths

    def _context(self, img):
        if self._ctx is None:
            from .context import VoContext
            h, w = img.shape
            self._ctx = VoContext(w, h, max_pts=64, device=self._device)
        return self._ctx

    def getImage(self, id):
        if id >= self._length or id < 0:
            raise AssertionError
        raw = imread_gray(self._image_paths[id])
        p = self._bilateral_filter_params
        return self._context(raw).bilateral(raw, p['d'], p['sigmaColor'], p['sigmaSpace'])

    def getPose(self, id):
        if id >= self._length or id < 0:
            raise AssertionError
        return self._poses[id]

    def getFrame(self, id):
        if id >= self._length or id < 0:
            raise AssertionError
        return self.getImage(id), self.getPose(id)

    def getCamera(self):
        return self._camera

    def getInit(self):
        """Returns Tuple with first and second index"""
        return tuple(self._cfg[self._name]['init'])
