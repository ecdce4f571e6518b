"""The .npy files between the stages (profiles/com_profs.npy, profiles/cov_profs.npy, latent.npy;
pipelines.py:315-321, ae_utils.py:299-325, cluster_utils.py:285-300) are written as the reference
writes them -- and the arrays are kept, so that the next stage of the SAME process does not read
gigabytes back from the file it has just written.  A cached array is handed out only while the
file on disk is the one this process wrote (size and mtime); callers must not modify it."""
import os

import numpy as np

_cache = {}


def _sig(path):
    st = os.stat(path)
    return (st.st_size, st.st_mtime_ns)


def _npy(path):
    return path if path.endswith(".npy") else path + ".npy"


def save(path, arr):
    """np.save(path, arr) and remember arr."""
    np.save(path, arr)
    p = os.path.abspath(_npy(path))
    _cache[p] = (_sig(p), arr)


def load(path):
    """np.load(path), from memory when this process wrote that very file."""
    p = os.path.abspath(_npy(path))
    hit = _cache.get(p)
    if hit is not None:
        try:
            if hit[0] == _sig(p):
                return hit[1]
        except OSError:
            pass
        del _cache[p]
    return np.load(p)


def drop(path=None):
    """Forget one array (or all)."""
    if path is None:
        _cache.clear()
    else:
        _cache.pop(os.path.abspath(_npy(path)), None)
