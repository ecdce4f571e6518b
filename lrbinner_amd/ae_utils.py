"""VAE stage -- drop-in for ``mbcclr_utils/ae_utils.py`` on PyTorch-ROCm.

Same entry point (``vae_encode(output, latent_dims, hidden_layers, epochs,
constraints, cuda)``), same inputs (``profiles/{com,cov}_profs.npy``), same
outputs (``model.pt`` with the reference's dict + state_dict keys, ``latent.npy``
float32), same architecture and loss (ae_utils.py:35-97,127-139,163-191,243-271).

What differs is how the GPU is fed.  The reference pushes 1024-row batches
through a DataLoader with worker processes (ae_utils.py:19-32); here the whole
scaled matrix lives in HBM (5 M x 168 float32 = 3.4 GB, nothing on a 288 GB
part), every epoch draws one device-side permutation, and loss bookkeeping
stays on the device until the epoch ends.  The dense layers are genuine GEMMs
and run on the matrix cores through rocBLAS/hipBLASLt.
"""
import json
import logging
import os

import numpy as np

from . import _npcache
import torch
from torch import nn, optim

logger = logging.getLogger('LRBinner')

with open(os.path.join(os.path.dirname(__file__), 'hyper_params.json')) as _f:
    h_params = json.load(_f)  # loss weights keyed by composition width (hyper_params.json:2-19)


def minmax_scale(x):
    """Column-wise [0,1] scaling with sklearn.MinMaxScaler's arithmetic
    (x * scale + min_, scale = 1/range, a constant column maps to 0)
    -- ae_utils.py:21-22."""
    x = np.asarray(x, dtype=np.float64)
    lo = x.min(axis=0)
    rng = x.max(axis=0) - lo
    rng[rng == 0.0] = 1.0
    scale = 1.0 / rng
    return x * scale + (0.0 - lo * scale)


def _minmax_scale_t(x):
    """minmax_scale on a float64 tensor, same operations in the same order (every one of them a
    single correctly rounded IEEE operation on either side: the results are the same bits)."""
    if x.shape[0] == 0:
        return x
    lo = x.amin(dim=0)
    rng = x.amax(dim=0) - lo
    rng[rng == 0.0] = 1.0
    scale = 1.0 / rng
    shift = 0.0 - lo * scale
    x = x * scale
    x += shift
    return x


def make_data(covs, profs, device):
    """Scaled float32 [N, cov+prof] matrix on ``device`` (coverage columns first,
    the order forward() concatenates them, ae_utils.py:185).  On a GPU the float64 profiles
    are uploaded and scaled there (seconds of host arithmetic at millions of reads)."""
    if torch.device(device).type == "cuda":
        out = torch.empty((covs.shape[0], covs.shape[1] + profs.shape[1]), dtype=torch.float32, device=device)
        c = covs.shape[1]
        for arr, lo_col in ((covs, 0), (profs, c)):
            t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(device)
            out[:, lo_col:lo_col + arr.shape[1]] = _minmax_scale_t(t)
            del t
        return out
    profs = minmax_scale(profs)
    covs = minmax_scale(covs)
    x = np.concatenate([covs, profs], axis=1).astype(np.float32)
    return torch.from_numpy(x).to(device)


class VAE(nn.Module):
    def __init__(self, cov_size, prof_size, *, latent_dims=8, hidden_layers=[128, 128],
                 constraints=None, device='cpu'):
        super().__init__()
        self.cov_size = cov_size
        self.prof_size = prof_size
        self.hidden_layers = list(hidden_layers)
        self.latent_dims = latent_dims
        self.dropout = 0.1

        # module names are the reference's: they are the state_dict keys of model.pt
        self.encoderlayers = nn.ModuleList()
        self.encodernorms = nn.ModuleList()
        self.decoderlayers = nn.ModuleList()
        self.decodernorms = nn.ModuleList()
        widths = [cov_size + prof_size] + self.hidden_layers
        for nin, nout in zip(widths[:-1], widths[1:]):
            self.encoderlayers.append(nn.Linear(nin, nout))
            self.encodernorms.append(nn.BatchNorm1d(nout))
        self.mu = nn.Linear(self.hidden_layers[-1], latent_dims)
        self.logsigma = nn.Linear(self.hidden_layers[-1], latent_dims)
        widths = [latent_dims] + self.hidden_layers[::-1]
        for nin, nout in zip(widths[:-1], widths[1:]):
            self.decoderlayers.append(nn.Linear(nin, nout))
            self.decodernorms.append(nn.BatchNorm1d(nout))
        self.outputlayer = nn.Linear(self.hidden_layers[0], cov_size + prof_size)

        self.relu = nn.LeakyReLU()
        self.softplus = nn.Softplus()
        self.dropoutlayer = nn.Dropout(p=self.dropout)

        self.constraints = constraints
        self.device = device
        self._ml = self._mnl = None
        if constraints:
            self._ml = self._pairs(constraints.get('ml', []))
            self._mnl = self._pairs(constraints.get('mnl', []))
        self.to(device)

    @staticmethod
    def _pairs(lst):
        """Unordered unique pairs as an int64 [P, 2] array (a pair listed in both
        orders counts once, as in the reference's i > j scan, ae_utils.py:100-125)."""
        s = {(min(int(a), int(b)), max(int(a), int(b))) for a, b in lst if int(a) != int(b)}
        return np.array(sorted(s), dtype=np.int64).reshape(-1, 2)

    # block = BatchNorm(Dropout(LeakyReLU(Linear(x))))  -- ae_utils.py:130-133,173-176
    def _encode(self, x):
        for layer, norm in zip(self.encoderlayers, self.encodernorms):
            x = norm(self.dropoutlayer(self.relu(layer(x))))
        return self.mu(x), self.softplus(self.logsigma(x))

    def _decode(self, z):
        for layer, norm in zip(self.decoderlayers, self.decodernorms):
            z = norm(self.dropoutlayer(self.relu(layer(z))))
        return self.outputlayer(z)

    def forward(self, x):
        mu, logsigma = self._encode(x)
        eps = torch.randn_like(mu)
        recon = self._decode(mu + eps * torch.exp(logsigma / 2))
        return recon, mu, logsigma

    def forward_predict(self, covs, profs):
        return self._encode(torch.cat((covs, profs), 1))

    def calc_loss(self, x, recon, mu, logsigma, indices=None):
        """ae_utils.py:243-271.  Returns (loss, e_cov, e_comp, kld)."""
        c = self.cov_size
        diff = (recon - x).pow(2)
        e_cov = diff[:, :c].sum(dim=1).mean()
        e_comp = diff[:, c:].sum(dim=1).mean()
        kld = -0.5 * (1 + logsigma - mu.pow(2) - logsigma.exp()).sum(dim=1).mean()
        w = h_params[str(self.prof_size)]
        loss = e_cov * w["e_cov_weight"] + e_comp * w["e_comp_weight"] + kld * w["kld_weight"]
        if self.constraints is not None and indices is not None and self._ml is not None \
                and len(self._ml) > 0:
            loss = loss + self._constraint_terms(mu, indices)
        return loss, e_cov, e_comp, kld

    def _constraint_terms(self, mu, indices):
        # position of every dataset row inside this batch (-1 = absent)
        pos = torch.full((self._n_rows,), -1, dtype=torch.long, device=mu.device)
        pos[indices] = torch.arange(len(indices), device=mu.device)

        def local(pairs):
            p = torch.from_numpy(pairs).to(mu.device)
            a, b = pos[p[:, 0]], pos[p[:, 1]]
            ok = (a >= 0) & (b >= 0)
            return a[ok], b[ok]

        a, b = local(self._ml)
        if len(a) == 0:  # both terms are gated on must-link pairs (ae_utils.py:250-253)
            return 0.0
        extra = (mu[a] - mu[b]).pow(2).sum(dim=1).mean()
        a, b = local(self._mnl)
        if len(a):
            extra = extra + torch.clamp(10 - (mu[a] - mu[b]).pow(2).sum(dim=1).mean(), min=0)
        return extra

    def _train_step(self, data, idx, optimizer, sums):
        """One optimisation step on rows ``idx`` (ae_utils.py:213-232)."""
        x = data.index_select(0, idx)
        recon, mu, logsigma = self(x)
        loss, e_cov, e_comp, kld = self.calc_loss(x, recon, mu, logsigma, idx)
        loss.backward()
        optimizer.step()
        sums += torch.stack([loss.detach(), e_cov.detach(), e_comp.detach(), kld.detach()])

    def trainmodel(self, data, *, nepochs=500, lrate=1e-3, batchsteps=(25, 75, 150, 300),
                   batch_size=1024, save_path=None, use_graph=None):
        """Adam over all parameters; the batch doubles at every epoch in ``batchsteps``;
        the ragged tail of each epoch is dropped (ae_utils.py:199-281).

        The step is ~45 k MACs per sample: on a GPU it is launch-bound (about thirty tiny
        kernels).  On CUDA/HIP devices the whole step -- gather, forward, loss, backward,
        Adam -- is therefore captured once per batch size in a HIP graph and replayed;
        only the 8 KB index vector changes between replays."""
        steps = set(batchsteps)
        self._n_rows = data.shape[0]
        on_gpu = data.is_cuda
        if on_gpu and use_graph is None and self.constraints is None \
                and os.environ.get("LRB_VAE_NATIVE", "1") != "0":
            return self._trainmodel_native(data, nepochs, lrate, steps, batch_size, save_path)
        if use_graph is None:
            use_graph = on_gpu and os.environ.get("LRB_VAE_GRAPH", "1") != "0" \
                and self.constraints is None
        optimizer = optim.Adam(self.parameters(), lr=lrate, capturable=bool(use_graph))
        n = data.shape[0]
        graphs = {}  # batch size -> (graph, static index buffer)
        for epoch in range(nepochs):
            if epoch in steps:
                batch_size *= 2
            self.train()
            nb = n // batch_size
            sums = self._sums = getattr(self, "_sums", None)
            if sums is None or sums.device != data.device:
                sums = self._sums = torch.zeros(4, device=data.device)
            sums.zero_()
            perm = torch.randperm(n, device=data.device)
            b = 0
            if use_graph and nb > 0 and batch_size not in graphs:
                # a few eager steps on a side stream warm the allocator (they are ordinary
                # training steps), then the step is recorded -- recording executes nothing
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    while b < min(3, nb):
                        optimizer.zero_grad(set_to_none=True)
                        self._train_step(data, perm[b * batch_size:(b + 1) * batch_size], optimizer, sums)
                        b += 1
                torch.cuda.current_stream().wait_stream(side)
                try:
                    static_idx = torch.zeros(batch_size, dtype=torch.long, device=data.device)
                    g = torch.cuda.CUDAGraph()
                    optimizer.zero_grad(set_to_none=True)
                    with torch.cuda.graph(g):
                        self._train_step(data, static_idx, optimizer, sums)
                    graphs[batch_size] = (g, static_idx)
                except Exception as e:  # capture unsupported: stay eager, say so
                    logger.debug(f"HIP graph capture of the VAE step failed ({e}); running eagerly")
                    use_graph = False
                    torch.cuda.synchronize()
            if use_graph and batch_size in graphs:
                g, static_idx = graphs[batch_size]
                while b < nb:
                    static_idx.copy_(perm[b * batch_size:(b + 1) * batch_size])
                    g.replay()
                    b += 1
            else:
                while b < nb:
                    optimizer.zero_grad(set_to_none=True)
                    self._train_step(data, perm[b * batch_size:(b + 1) * batch_size], optimizer, sums)
                    b += 1
            if logger.isEnabledFor(logging.DEBUG):
                s = (sums / (1 + nb)).tolist()
                logger.debug(f'Epoch: {epoch + 1:4} Loss: {s[0]:.6f}\tEC: {s[1]:.7f}\t'
                             f'EP: {s[2]:.6f}\tKLD: {s[3]:.4f}\tBatchsize: {batch_size}')
        self._sums = None
        if save_path is not None:
            self.save(save_path)

    def _trainmodel_native(self, data, nepochs, lrate, steps, batch_size, save_path):
        """The same schedule on the fused HIP step (include/lrb_hip.h K7, csrc/lrb_vae.hip):
        12 kernels per step in a hipGraph instead of ~190 autograd kernels.  Parameters,
        running statistics and num_batches_tracked come back into this module, so model.pt
        and encode() are unchanged.  Dropout masks and eps come from the library's
        counter-based generator, seeded from torch's (the reference is unseeded)."""
        from . import device as lrb
        from .vae_native import NativeTrainer
        n = data.shape[0]
        max_batch = batch_size * (2 ** sum(1 for e in steps if e < nepochs))
        w = h_params[str(self.prof_size)]
        dev_index = data.device.index if data.device.index is not None else torch.cuda.current_device()
        ctx = lrb.Context(dev_index, use_torch_stream=True)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        tr = NativeTrainer(ctx, self, max_batch=max(max_batch, 2),
                           loss_weights=[w["e_cov_weight"], w["e_comp_weight"], w["kld_weight"]],
                           lr=lrate, seed=seed)
        data = data.contiguous()
        try:
            tr.push()
            perm = torch.empty(n, dtype=torch.long, device=data.device)  # one buffer: the recorded step points at it
            for epoch in range(nepochs):
                if epoch in steps:
                    batch_size *= 2
                nb = n // batch_size
                torch.randperm(n, device=data.device, out=perm)
                debug = logger.isEnabledFor(logging.DEBUG)
                if debug:
                    tr.zero_sums()
                tr.train(data, perm, batch_size, nb)
                if debug:
                    s = (tr.sums() / (1 + nb)).tolist()
                    logger.debug(f'Epoch: {epoch + 1:4} Loss: {s[0]:.6f}\tEC: {s[1]:.7f}\t'
                                 f'EP: {s[2]:.6f}\tKLD: {s[3]:.4f}\tBatchsize: {batch_size}')
            torch.cuda.current_stream().synchronize()
            tr.pull()
        except BaseException:
            tr.close()
            ctx.close()
            raise
        # kept for encode(): same parameters and running statistics as the module now has
        self.release_native()
        self._native = (tr, ctx)
        if save_path is not None:
            self.save(save_path)

    def release_native(self):
        """Free the fused trainer left behind by the last training run (if any)."""
        nat = getattr(self, "_native", None)
        if nat is not None:
            nat[0].close()
            nat[1].close()
            self._native = None

    @torch.no_grad()
    def encode(self, data, chunk=1 << 18):
        """Latent means (eval mode: running statistics, no dropout), float32, input
        order -- ae_utils.py:141-161."""
        nat = getattr(self, "_native", None)
        if nat is not None and data.is_cuda and os.environ.get("LRB_VAE_NATIVE", "1") != "0":
            # the trainer still holds these parameters: three fused kernels per 8 k rows
            mu = nat[0].encode(data.contiguous())
            torch.cuda.current_stream().synchronize()
            return mu.cpu().numpy()
        self.eval()
        out = np.empty((data.shape[0], self.latent_dims), dtype=np.float32)
        for s in range(0, data.shape[0], chunk):
            mu, _ = self._encode(data[s:s + chunk])
            out[s:s + chunk] = mu.float().cpu().numpy()
        return out

    def save(self, save_path):
        torch.save({'cov_size': self.cov_size, 'prof_size': self.prof_size,
                    'dropout': self.dropout, 'hidden_layers': self.hidden_layers,
                    'latent_dims': self.latent_dims, 'state': self.state_dict()}, save_path)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def vae_encode(output, latent_dims, hidden_layers, epochs, constraints, cuda):
    import time
    t0 = time.time()
    comp_profiles = _npcache.load(f"{output}/profiles/com_profs.npy")
    cov_profiles = _npcache.load(f"{output}/profiles/cov_profs.npy")
    device = "cuda" if cuda else "cpu"

    vae = VAE(cov_profiles.shape[1], comp_profiles.shape[1], latent_dims=latent_dims,
              hidden_layers=hidden_layers, constraints=constraints, device=device)
    logger.debug(f"Model param count = {count_parameters(vae)}")
    logger.debug(vae)

    data = make_data(cov_profiles, comp_profiles, device)
    t1 = time.time()
    vae.trainmodel(data, save_path=f"{output}/model.pt", nepochs=epochs, batchsteps=[50, 100, 150])
    t2 = time.time()
    latent = vae.encode(data)
    vae.release_native()
    _npcache.save(f"{output}/latent", latent)
    logger.debug(f"VAE stage: load + scale + upload {t1 - t0:.2f} s, training {t2 - t1:.2f} s, "
                 f"encode + latent.npy {time.time() - t2:.2f} s")
