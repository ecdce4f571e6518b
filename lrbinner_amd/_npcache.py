"""The .npy files between the stages (profiles/com_profs.npy, profiles/cov_profs.npy, latent.npy;
pipelines.py:315-321, ae_utils.py:299-325, cluster_utils.py:285-300) are written as the reference
writes them -- and the arrays are kept, so that the next stage of the SAME process does not read
gigabytes back from the file it has just written.  A cached array is handed out only while the
file on disk is the one this process wrote (size and mtime); callers must not modify it."""
import os
import threading

import numpy as np

_cache = {}
_pending = {}  # abs path -> (thread, [exception]) of a save still on its way to the disk


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


def save_async(path, arr):
    """np.save(path, arr) on a thread of its own; the array is handed out by load() from now on.  The file appears
    under its name only when complete (written beside it, then renamed): a process that dies first leaves no
    half-written .npy behind, and the stage that owns the file runs again on --resume (pipelines._stage, late
    artifacts).  finish() waits and reports a failed write."""
    p = os.path.abspath(_npy(path))
    finish(p)
    if os.path.exists(p):
        os.remove(p)
    err = []

    def work():
        tmp = p + ".part"
        try:
            with open(tmp, "wb") as f:
                np.save(f, arr)
            os.replace(tmp, p)
        except BaseException as e:  # reported by finish() on the caller's thread
            err.append(e)
            try:
                os.remove(tmp)
            except OSError:
                pass

    th = threading.Thread(target=work, daemon=True)
    _cache[p] = (None, arr)
    _pending[p] = (th, err)
    th.start()


def finish(path=None):
    """Wait for the saves still in flight (one file, or all); raises what a failed write raised."""
    keys = [os.path.abspath(_npy(path))] if path is not None else list(_pending)
    for p in keys:
        job = _pending.pop(p, None)
        if job is None:
            continue
        job[0].join()
        if job[1]:
            _cache.pop(p, None)
            raise job[1][0]
        if p in _cache and _cache[p][0] is None:
            _cache[p] = (_sig(p), _cache[p][1])


def load(path):
    """np.load(path), from memory when this process wrote that very file."""
    p = os.path.abspath(_npy(path))
    hit = _cache.get(p)
    if hit is not None and hit[0] is None and p in _pending:
        return hit[1]                      # still being written by this process: the array itself
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
    finish(path)
    if path is None:
        _cache.clear()
    else:
        _cache.pop(os.path.abspath(_npy(path)), None)
