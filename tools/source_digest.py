"""Digest of kernel sources as bench.py / tools/summarize_profiles.py stamp profile constants with it: comments and white space do not count
(a committed rocprofv3 summary stays valid across comment edits), every token of code does."""
import hashlib
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visual-odom-pipeline_amd", "csrc")
_COMMENT = re.compile(r"/\*.*?\*/|//[^\n]*", re.S)


def code_bytes(path):
    with open(path) as f:
        txt = f.read()
    return re.sub(r"\s+", " ", _COMMENT.sub(" ", txt)).strip().encode()


def digest(names=None):
    """names: files under csrc/ (default: every .hip / .h, sorted)"""
    if names is None:
        names = sorted(n for n in os.listdir(CSRC) if n.endswith((".hip", ".h")))
    h = hashlib.sha256()
    for n in names:
        h.update(code_bytes(os.path.join(CSRC, n)))
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print("csrc", digest(), "klt", digest(["vo_klt.hip"]), "klt+frame", digest(["vo_klt.hip", "vo_frame.hip"]))
