"""Training of the two recurrent networks of the path in PyTorch (ROCm), and export of the
result to the weight bundles the HIP kernels load.

Reference: training/barcodes/scripts/train_demux_nn.py:81-101 (demux classifier, weighted
categorical cross-entropy, early stopping, best-model checkpoints) and
training/signal-scaling/scripts/learn-scaling.py:35-56 (scaler regressor, MSE).  There the
models are Keras graphs trained under tf.distribute.MirroredStrategy; here:

  * the modules are the architectures of the SHIPPED model files (demux-tetra-r4:
    GaussianNoise -> Bidirectional LSTM(48) -> Dropout -> LSTM(64) -> Dropout -> Dense(5);
    scaler-r3: GaussianNoise -> LSTM(48, sequences) -> Dropout -> LSTM(48) -> Dropout ->
    Dense(2)) because those shapes are what csrc/k_lstm.hip keeps in registers -- the scaler of
    the training script (bidirectional + BatchNorm) is not what the reference ships;
  * data parallelism is one process per GPU: ``torch.nn.parallel.DistributedDataParallel``
    over RCCL (gloo in the CPU tests), every rank a contiguous shard of each epoch's
    permutation, gradients all-reduced in buckets during backward;
  * ``export_demux_bundle`` / ``export_scaler_bundle`` write the .npz layout of
    ``config.load_model_arrays`` (Keras kernel / recurrent_kernel / bias, gate order
    i, f, c, o), so a trained model drops into ``pxg_create`` unchanged; the demux bundle
    carries a calibration table (phred -> minimal score) fitted on held-out predictions.

No real training data exists in this repository: ``synthetic_demux_dataset`` /
``synthetic_scaler_dataset`` draw from the same generator as the bench reads.
"""
import csv
import os

import numpy as np
import torch
from torch import nn

__all__ = ['DemuxClassifier', 'ScalerRegressor', 'Trainer', 'weighted_cross_entropy',
           'weighted_accuracy', 'fit_calibration', 'export_demux_bundle', 'export_scaler_bundle',
           'synthetic_demux_dataset', 'synthetic_scaler_dataset']


class _KerasLSTM(nn.LSTM):
    """nn.LSTM whose second bias stays zero: Keras has ONE bias per gate, and the export must
    be exact (kernel = weight_ih^T, recurrent_kernel = weight_hh^T, bias = bias_ih)."""

    def __init__(self, inputs, units):
        super().__init__(inputs, units, batch_first=True)
        with torch.no_grad():
            self.bias_hh_l0.zero_()
            self.bias_ih_l0[units:2 * units].fill_(1.0)        # unit_forget_bias
        self.bias_hh_l0.requires_grad_(False)

    def keras_arrays(self):
        return (self.weight_ih_l0.detach().cpu().numpy().T.copy(),
                self.weight_hh_l0.detach().cpu().numpy().T.copy(),
                self.bias_ih_l0.detach().cpu().numpy().copy())


class DemuxClassifier(nn.Module):
    """[B, 300] normalised adapter windows -> log-probabilities [B, n_classes]."""

    def __init__(self, n_classes=5, noise=0.05, drop1=0.2, drop2=0.3):
        super().__init__()
        self.noise = noise
        self.fwd, self.bwd = _KerasLSTM(1, 48), _KerasLSTM(1, 48)
        self.top = _KerasLSTM(96, 64)
        self.drop1, self.drop2 = nn.Dropout(drop1), nn.Dropout(drop2)
        self.dense = nn.Linear(64, n_classes)

    def forward(self, x):
        x = x.unsqueeze(-1)
        if self.training and self.noise:
            x = x + self.noise * torch.randn_like(x)
        hf, _ = self.fwd(x)
        hb, _ = self.bwd(torch.flip(x, dims=[1]))
        h = self.drop1(torch.cat([hf, torch.flip(hb, dims=[1])], dim=-1))
        _, (hn, _) = self.top(h)
        return torch.log_softmax(self.dense(self.drop2(hn[0])), dim=-1)


class ScalerRegressor(nn.Module):
    """[B, 2000] pooled read heads -> standardised (scale, shift) [B, 2]."""

    def __init__(self, noise=1.5, drop1=0.1, drop2=0.2):
        super().__init__()
        self.noise = noise
        self.l1, self.l2 = _KerasLSTM(1, 48), _KerasLSTM(48, 48)
        self.drop1, self.drop2 = nn.Dropout(drop1), nn.Dropout(drop2)
        self.dense = nn.Linear(48, 2)

    def forward(self, x):
        x = x.unsqueeze(-1)
        if self.training and self.noise:
            x = x + self.noise * torch.randn_like(x)
        h, _ = self.l1(x)
        _, (hn, _) = self.l2(self.drop1(h))
        return self.dense(self.drop2(hn[0]))


# ---- objective of the classifier (weighted_metrics.py; train_demux_nn.py:126-133) ---------
def cost_matrix(n_classes, cross_contamination_penalty=2.0):
    """Calling one barcode for another costs more than mixing up decoy and barcode."""
    m = torch.ones(n_classes, n_classes)
    m[1:, 1:] *= cross_contamination_penalty
    return m


def _sample_weights(logp, labels, cost):
    return cost.to(logp.device)[labels, logp.argmax(dim=1)]


def weighted_cross_entropy(logp, labels, cost, class_weight=None):
    """Cross-entropy of every sample times cost[true class, currently predicted class] (and
    an optional per-class weight), averaged over the batch."""
    w = _sample_weights(logp, labels, cost)
    if class_weight is not None:
        w = w * class_weight.to(logp.device)[labels]
    return (w * nn.functional.nll_loss(logp, labels, reduction='none')).mean()


def weighted_accuracy(logp, labels, cost):
    w = _sample_weights(logp, labels, cost)
    return float((w * (logp.argmax(dim=1) == labels)).sum() / w.sum())


# ---- trainer -----------------------------------------------------------------------------
class Trainer:
    """Adam + early stopping on the validation loss + best-model snapshot + CSV log.

    `dist`: an initialised torch.distributed module for data parallelism (one process per
    GPU); the model is wrapped in DistributedDataParallel, every rank trains on its shard of
    each epoch's permutation (same seed everywhere), the validation loss is all-reduced so
    every rank takes the same early-stopping decision; rank 0 writes the artefacts."""

    def __init__(self, model, loss_fn, device='cpu', dist=None, lr=1e-3, batch_size=256,
                 epochs=20, validation_split=0.1, patience=5, min_delta=1e-4, seed=922,
                 output_dir=None):
        self.device, self.dist = torch.device(device), dist
        self.model = model.to(self.device)
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.net = self.model
        if dist is not None:
            self.net = nn.parallel.DistributedDataParallel(
                self.model, device_ids=[self.device.index] if self.device.type == 'cuda' else None)
        self.loss_fn = loss_fn
        self.opt = torch.optim.Adam([p for p in self.model.parameters() if p.requires_grad], lr=lr)
        self.batch_size, self.epochs = batch_size, epochs
        self.validation_split, self.patience, self.min_delta = validation_split, patience, min_delta
        self.seed, self.output_dir = seed, output_dir
        self.history = []

    def _mean_over_ranks(self, total, count):
        t = torch.tensor([total, count], dtype=torch.float64, device=self.device)
        if self.dist is not None:
            self.dist.all_reduce(t)
        return float(t[0] / t[1].clamp(min=1))

    def evaluate(self, x, y):
        self.net.eval()
        total, count = 0.0, 0
        with torch.no_grad():
            for a in range(self.rank * self.batch_size, len(x), self.batch_size * self.world):
                xb, yb = x[a:a + self.batch_size].to(self.device), y[a:a + self.batch_size].to(self.device)
                total += float(self.loss_fn(self.net(xb), yb)) * len(xb)
                count += len(xb)
        return self._mean_over_ranks(total, count)

    def fit(self, x, y):
        g = torch.Generator().manual_seed(self.seed)
        order = torch.randperm(len(x), generator=g)
        n_val = int(len(x) * self.validation_split)
        val, train = order[:n_val], order[n_val:]
        best, best_state, stale = None, None, 0
        for epoch in range(self.epochs):
            self.net.train()
            perm = train[torch.randperm(len(train), generator=g)]
            per_rank = (len(perm) + self.world - 1) // self.world
            if len(perm) == 0:
                raise ValueError('no training samples left after the validation split')
            # every rank gets exactly per_rank samples: the permutation wraps around (what
            # DistributedSampler does), so no rank ever steps on an empty or shorter batch -- an
            # empty batch is a NaN loss that DDP all-reduces into every rank's weights
            wrapped = perm[torch.arange(self.world * per_rank) % len(perm)]
            mine = wrapped[self.rank * per_rank:(self.rank + 1) * per_rank]
            steps = (per_rank + self.batch_size - 1) // self.batch_size      # same on every rank
            total, count = 0.0, 0
            for k in range(steps):
                idx = mine[k * self.batch_size:(k + 1) * self.batch_size]
                xb, yb = x[idx].to(self.device), y[idx].to(self.device)
                self.opt.zero_grad(set_to_none=True)
                loss = self.loss_fn(self.net(xb), yb)
                loss.backward()                        # DDP all-reduces the gradient buckets here
                self.opt.step()
                total += float(loss) * len(idx)
                count += len(idx)
            row = {'epoch': epoch, 'loss': self._mean_over_ranks(total, count),
                   'val_loss': self.evaluate(x[val], y[val]) if n_val else float('nan')}
            self.history.append(row)
            monitored = row['val_loss'] if n_val else row['loss']
            if best is None or monitored < best - self.min_delta:
                best, stale = monitored, 0
                best_state = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
            else:
                stale += 1
                if stale >= self.patience:
                    break
        if best_state is not None:
            self.model.load_state_dict(best_state)
        if self.output_dir and self.rank == 0:
            os.makedirs(self.output_dir, exist_ok=True)
            with open(os.path.join(self.output_dir, 'training-log.csv'), 'w', newline='') as fh:
                w = csv.DictWriter(fh, fieldnames=['epoch', 'loss', 'val_loss'])
                w.writeheader()
                w.writerows(self.history)
            torch.save(self.model.state_dict(), os.path.join(self.output_dir, 'bestmodel-validation.pt'))
        return self.history


# ---- export to the bundles the GPU library loads ------------------------------------------
def fit_calibration(scores, correct, n_rows=29):
    """phred q -> smallest score s such that calls with score >= s are right with probability
    >= 1 - 10^(-q/10) on held-out data (row 0 = 0.0; monotone; same shape as the shipped
    poreplex_params/calibration table that barcoding.py:55-60 bisects)."""
    scores, correct = np.asarray(scores, np.float64), np.asarray(correct, bool)
    order = np.argsort(-scores)
    s, ok = scores[order], np.cumsum(correct[order])
    precision = ok / np.arange(1, len(s) + 1)
    table = np.zeros(n_rows, dtype=np.float64)
    for q in range(1, n_rows):
        good = np.nonzero(precision >= 1.0 - 10.0 ** (-q / 10.0))[0]
        lowest = s[good[-1]] if len(good) else 1.0
        table[q] = max(table[q - 1] + 1e-9, min(float(lowest), 1.0 - 1e-7 * (n_rows - q)))
    return table


def export_demux_bundle(model, path, calibration):
    arrays = {}
    for name, layer in (('fwd', model.fwd), ('bwd', model.bwd), ('top', model.top)):
        k, r, b = layer.keras_arrays()
        arrays[name + '_kernel'], arrays[name + '_recurrent'], arrays[name + '_bias'] = k, r, b
    arrays['dense_kernel'] = model.dense.weight.detach().cpu().numpy().T.copy()
    arrays['dense_bias'] = model.dense.bias.detach().cpu().numpy().copy()
    arrays['calibration'] = np.asarray(calibration, dtype=np.float64)
    np.savez(path, **{k: np.ascontiguousarray(v) for k, v in arrays.items()})
    return path


def export_scaler_bundle(model, path, output_transform, stride=15, length=30000, min_length=9000):
    arrays = {}
    for name, layer in (('lstm1', model.l1), ('lstm2', model.l2)):
        k, r, b = layer.keras_arrays()
        arrays[name + '_kernel'], arrays[name + '_recurrent'], arrays[name + '_bias'] = k, r, b
    arrays['dense_kernel'] = model.dense.weight.detach().cpu().numpy().T.copy()
    arrays['dense_bias'] = model.dense.bias.detach().cpu().numpy().copy()
    arrays['output_transform'] = np.asarray(output_transform, dtype=np.float64)   # scale mean/std, shift mean/std
    arrays['input_stride'], arrays['input_length'] = np.int64(stride), np.int64(length)
    arrays['input_min_length'] = np.int64(min_length)
    np.savez(path, **{k: (np.ascontiguousarray(v) if np.ndim(v) else np.asarray(v)) for k, v in arrays.items()})
    return path


# ---- synthetic training sets (same generator as the bench reads) --------------------------
def synthetic_demux_dataset(n, seed=922, noise=0.35):
    """(windows [n, 300] float32, labels [n] int64): class 0 = decoy (plain adapter noise),
    classes 1..4 = the synthetic barcode prototypes with per-read distortion."""
    from .synth import load_prototypes
    rng = np.random.Generator(np.random.PCG64(seed))
    protos = load_prototypes()
    labels = rng.integers(0, 5, n)
    x = rng.standard_normal((n, 300)).astype(np.float32)
    for c in range(1, 5):
        m = labels == c
        gain = rng.uniform(0.8, 1.2, (int(m.sum()), 1)).astype(np.float32)
        x[m] = gain * protos[c] + noise * x[m]
    return torch.from_numpy(x), torch.from_numpy(labels.astype(np.int64))


def synthetic_scaler_dataset(n, seed=922, samples=32000):
    """(heads [n, 2000] float32, standardised (scale, shift) [n, 2], output_transform):
    pooled pA heads of synthetic reads and the scaling that maps them onto the model levels."""
    from .synth import synth_batch
    sb = synth_batch(n, seed=seed, samples_per_read=samples, jitter=0.02)
    o, cal = sb['offsets'], sb['calib']
    heads = np.zeros((n, 2000), dtype=np.float32)
    for i in range(n):
        raw = sb['arena'][o[i]:o[i + 1]][:30000].astype(np.float64)
        pa = ((raw + cal['offset'][i]) * (cal['range'][i] / cal['digitisation'][i])).astype(np.float32)
        m = pa[:len(pa) - len(pa) % 15].reshape(-1, 15).mean(axis=1, dtype=np.float32)
        heads[i, 2000 - len(m):] = m
    ss = sb['scale_shift'].astype(np.float64)
    xfrm = [ss[:, 0].mean(), ss[:, 0].std(), ss[:, 1].mean(), ss[:, 1].std()]
    target = np.stack([(ss[:, 0] - xfrm[0]) / xfrm[1], (ss[:, 1] - xfrm[2]) / xfrm[3]], axis=1)
    return torch.from_numpy(heads), torch.from_numpy(target.astype(np.float32)), xfrm
