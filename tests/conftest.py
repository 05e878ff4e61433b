import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# NB: oracle/ref_stub (our stand-in `cv2` package) is deliberately NOT on the path: a real OpenCV on the box must stay importable
# as `cv2` (tests/test_gpu_live_cv2.py); tests that need the stub's Rodrigues load it by file (helpers.ref_stub_cv2)
for p in (os.path.join(ROOT, "visual-odom-pipeline_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def seq3():
    """three 1241x376 synthetic frames + their motions"""
    from vo_mi355x import synthetic as syn
    return syn.make_sequence(3)


@pytest.fixture(scope="session")
def seq_small():
    """three 320x240 frames: pyramid truncates at level 2 (40x30 would be <= 31)"""
    from vo_mi355x import synthetic as syn
    return syn.make_sequence(3, w=320, h=240, seed=77, margin=64)
