"""Python face of the fused VAE training step (include/lrb_hip.h K7, csrc/lrb_vae.hip).

``NativeTrainer`` owns an ``lrb_vae`` object, moves the parameters of an ``ae_utils.VAE``
module into its flat vectors and back (same state_dict afterwards, running statistics and
``num_batches_tracked`` included), and runs epochs on a data matrix resident in HBM."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import call, vp

F32P = C.POINTER(C.c_float)


def _fp(a):
    return a.ctypes.data_as(F32P)


class NativeTrainer:
    def __init__(self, ctx, vae, max_batch, loss_weights, lr=1e-3, seed=0):
        self.ctx, self.vae = ctx, vae
        hidden = (C.c_int * len(vae.hidden_layers))(*vae.hidden_layers)
        w = (C.c_float * 3)(*[float(x) for x in loss_weights])
        self._h = vp()
        call("lrb_vae_create", ctx._h, int(vae.cov_size), int(vae.prof_size), hidden, len(vae.hidden_layers),
             int(vae.latent_dims), int(max_batch), w, float(lr), float(vae.dropout), int(seed) & (2 ** 64 - 1),
             C.byref(self._h))
        n, r = C.c_uint64(0), C.c_uint64(0)
        call("lrb_vae_sizes", self._h, C.byref(n), C.byref(r))
        self.n_params, self.n_running = n.value, r.value

    # ---- flat vector <-> module -------------------------------------------------------
    def _param_tensors(self):
        """The module's tensors in the order of the flat parameter vector."""
        v = self.vae
        out = []
        for lin, bn in zip(v.encoderlayers, v.encodernorms):
            out += [lin.weight, lin.bias, bn.weight, bn.bias]
        out += [(v.mu.weight, v.logsigma.weight), (v.mu.bias, v.logsigma.bias)]
        for lin, bn in zip(v.decoderlayers, v.decodernorms):
            out += [lin.weight, lin.bias, bn.weight, bn.bias]
        out += [v.outputlayer.weight, v.outputlayer.bias]
        return out

    def _norms(self):
        return list(self.vae.encodernorms) + list(self.vae.decodernorms)

    def push(self):
        """module -> trainer (parameters and running statistics)."""
        import torch
        parts = []
        for t in self._param_tensors():
            ts = t if isinstance(t, tuple) else (t,)
            parts += [x.detach().float().cpu().numpy().ravel() for x in ts]
        flat = np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)
        assert flat.size == self.n_params, (flat.size, self.n_params)
        call("lrb_vae_set", self._h, 0, _fp(flat), flat.size)
        run = np.concatenate([np.concatenate([bn.running_mean.detach().cpu().numpy(),
                                              bn.running_var.detach().cpu().numpy()]) for bn in self._norms()])
        run = np.ascontiguousarray(run, dtype=np.float32)
        call("lrb_vae_set", self._h, 1, _fp(run), run.size)

    def get(self, what, count):
        out = np.empty(count, np.float32)
        call("lrb_vae_get", self._h, int(what), _fp(out), count)
        return out

    def pull(self):
        """trainer -> module."""
        import torch
        flat = self.get(0, self.n_params)
        off = 0
        with torch.no_grad():
            for t in self._param_tensors():
                for x in (t if isinstance(t, tuple) else (t,)):
                    n = x.numel()
                    x.copy_(torch.from_numpy(flat[off:off + n].reshape(tuple(x.shape))).to(x.device))
                    off += n
            run = self.get(1, self.n_running)
            steps = self.steps_done()
            off = 0
            for bn in self._norms():
                n = bn.running_mean.numel()
                bn.running_mean.copy_(torch.from_numpy(run[off:off + n]).to(bn.running_mean.device))
                bn.running_var.copy_(torch.from_numpy(run[off + n:off + 2 * n]).to(bn.running_var.device))
                bn.num_batches_tracked.fill_(steps)
                off += 2 * n

    def steps_done(self):
        s = C.c_uint64(0)
        call("lrb_vae_steps_done", self._h, C.byref(s))
        return s.value

    # ---- training ----------------------------------------------------------------------
    def zero_sums(self):
        z = np.zeros(4, np.float32)
        call("lrb_vae_set", self._h, 4, _fp(z), 4)

    def sums(self):
        return self.get(4, 4)

    def train(self, data_t, perm_t, batch_size, n_steps, use_graph=True):
        """n_steps steps over consecutive slices of perm_t (int64 CUDA tensor of row ids)."""
        call("lrb_vae_train_dev", self._h, vp(data_t.data_ptr()), vp(perm_t.data_ptr()), int(batch_size),
             int(n_steps), 1 if use_graph else 0)

    def encode(self, data_t):
        """mu of every row (eval mode) as a CUDA float32 tensor [n][latent]; uses the parameters
        and running statistics currently in the trainer."""
        import torch
        n = data_t.shape[0]
        out = torch.empty((max(n, 1), int(self.vae.latent_dims)), dtype=torch.float32, device=data_t.device)
        call("lrb_vae_encode_dev", self._h, vp(data_t.data_ptr()), int(n), vp(out.data_ptr()))
        return out[:n]

    def debug(self, which, count):
        out = np.empty(count, np.float32)
        call("lrb_vae_debug_read", self._h, int(which), _fp(out), count)
        return out

    def close(self):
        if self._h:
            _lib.lib().lrb_vae_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hash_u32(seed, step, stream, idx):
    """The kernels' counter-based generator (csrc/lrb_vae.hip vae_hash), vectorised."""
    idx = np.asarray(idx, dtype=np.uint64)
    M = np.uint64(0xFFFFFFFF)
    h = (np.uint64(seed) ^ (np.uint64(step) * np.uint64(0x9E3779B9) & M) ^ (np.uint64(stream) * np.uint64(0x85EBCA6B) & M)) & M
    x = (idx * np.uint64(0x9E3779B1) + h) & M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & M
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & M
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def seed32(seed):
    seed = int(seed) & (2 ** 64 - 1)
    return (seed ^ (seed >> 32)) & 0xFFFFFFFF


def keep_mask(seed, step, stream, B, N, p):
    """Dropout keep mask [B][N] of one block at one step."""
    thr = np.uint32(int(p * 4294967296.0))
    return (hash_u32(seed32(seed), step, stream, np.arange(B * N)) >= thr).reshape(B, N)


def eps_normal(seed, step, B, L):
    """The eps [B][L] the heads kernel draws at one step (csrc/lrb_vae.hip vae_normal, stream 100):
    Box-Muller on two hashed uniforms, in float32 like the kernel (libm vs device log/cos may differ
    in the last bit)."""
    idx = np.arange(B * L, dtype=np.uint64)
    a = hash_u32(seed32(seed), step, 100, 2 * idx)
    b = hash_u32(seed32(seed), step, 100, 2 * idx + 1)
    scale = np.float32(2.3283064365386963e-10)
    u1 = (a.astype(np.float32) + np.float32(1.0)) * scale
    u2 = b.astype(np.float32) * scale
    e = np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)
    return e.astype(np.float32).reshape(B, L)
