"""poreplex_amd/training.py: the two networks trained in PyTorch and exported to the weight
bundles the GPU library loads.  CPU legs: tiny runs (loss goes down, early stopping, export ->
oracle forward == torch forward), 2-rank DDP over gloo.  -m gpu leg: training on the GPU and
the exported bundles through the HIP kernels."""
import copy
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

torch = pytest.importorskip('torch')
torch.set_num_threads(2)          # see conftest.py: no thread pool per core next to the 2-rank helpers

from poreplex_amd import training as TR  # noqa: E402
from poreplex_amd.config import default_config  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def config_with(demux=None, scaler=None):
    cfg = default_config()
    cfg = copy.deepcopy(cfg)
    if demux:
        cfg['demultiplexing']['demux_model'] = demux
    if scaler:
        cfg['signal_processing']['scaler_model'] = scaler
    return cfg


def train_small_demux(tmp_path, device='cpu', n=384, epochs=4):
    torch.manual_seed(0)
    x, y = TR.synthetic_demux_dataset(n, seed=5)
    model = TR.DemuxClassifier()
    cost = TR.cost_matrix(5)
    tr = TR.Trainer(model, lambda lp, t: TR.weighted_cross_entropy(lp, t, cost), device=device,
                    batch_size=64, epochs=epochs, lr=5e-3, patience=3, output_dir=str(tmp_path / 'demux'))
    hist = tr.fit(x, y)
    return model, hist, (x, y), cost


def test_demux_training_learns_and_exports(tmp_path):
    from oracle.pxo import Oracle
    model, hist, (x, y), cost = train_small_demux(tmp_path)
    assert hist[-1]['loss'] < hist[0]['loss'] and np.isfinite(hist[-1]['val_loss'])
    assert (tmp_path / 'demux' / 'training-log.csv').read_text().count('\n') == len(hist) + 1
    model.eval()
    with torch.no_grad():
        logp = model(x)
    assert TR.weighted_accuracy(logp, y, cost) > 0.3
    table = TR.fit_calibration(logp.exp().max(1).values.numpy(), (logp.argmax(1) == y).numpy())
    assert len(table) == 29 and table[0] == 0.0 and (np.diff(table) > 0).all() and table[-1] < 1.0
    path = TR.export_demux_bundle(model, str(tmp_path / 'demux-trained.npz'), table)
    orc = Oracle(config_with(demux=path))
    got = np.stack([orc.demux_forward(w) for w in x[:12].numpy()])
    assert np.abs(got - logp[:12].exp().numpy()).max() < 1e-4        # north_star tolerance


def test_scaler_training_and_export(tmp_path):
    from oracle.pxo import Oracle
    torch.manual_seed(1)
    x, y, xfrm = TR.synthetic_scaler_dataset(48, seed=9, samples=20000)
    x = x - 90.0                      # centre the pA levels: keeps the untrained LSTM out of saturation
    model = TR.ScalerRegressor(noise=0.5)
    tr = TR.Trainer(model, torch.nn.functional.mse_loss, batch_size=16, epochs=2, lr=2e-3,
                    validation_split=0.0, output_dir=str(tmp_path / 'scaler'))
    hist = tr.fit(x, y)
    assert np.isfinite(hist[-1]['loss'])
    path = TR.export_scaler_bundle(model, str(tmp_path / 'scaler-trained.npz'), xfrm)
    orc = Oracle(config_with(scaler=path))
    model.eval()
    with torch.no_grad():
        want = model(x[:4]).numpy()
    got = np.stack([orc.scaler_forward(h) for h in x[:4].numpy()])
    assert np.abs(got - want).max() < 2e-4


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np, torch
    import torch.distributed as dist
    from poreplex_amd import training as TR
    dist.init_process_group('gloo')
    torch.manual_seed(0)                       # same initial weights on both ranks (DDP broadcasts anyway)
    x, y = TR.synthetic_demux_dataset(256, seed=5)
    cost = TR.cost_matrix(5)
    tr = TR.Trainer(TR.DemuxClassifier(), lambda lp, t: TR.weighted_cross_entropy(lp, t, cost),
                    dist=dist, batch_size=32, epochs=2, lr=5e-3, output_dir={out!r})
    hist = tr.fit(x, y)
    flat = torch.cat([p.detach().flatten() for p in tr.model.parameters()])
    both = [torch.zeros_like(flat) for _ in range(2)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), 'ranks diverged'
    assert np.isfinite(hist[-1]['val_loss'])
    dist.barrier()
    open(os.path.join({out!r}, 'rank%d.ok' % dist.get_rank()), 'w').write('%.6f' % hist[-1]['loss'])
    dist.destroy_process_group()
""")


def test_two_rank_data_parallel_training_keeps_ranks_identical(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
        capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert (tmp_path / 'rank0.ok').read_text() == (tmp_path / 'rank1.ok').read_text()
    assert (tmp_path / 'training-log.csv').exists()


@pytest.mark.gpu
def test_train_on_gpu_and_run_the_exported_bundles_through_the_kernels(tmp_path):
    from poreplex_amd.native import NativeContext
    model, hist, (x, y), cost = train_small_demux(tmp_path, device='cuda', n=1024, epochs=3)
    assert hist[-1]['loss'] < hist[0]['loss']
    model.eval()
    with torch.no_grad():
        logp = model(x.cuda()).cpu()
    table = TR.fit_calibration(logp.exp().max(1).values.numpy(), (logp.argmax(1) == y).numpy())
    dpath = TR.export_demux_bundle(model.cpu(), str(tmp_path / 'demux-trained.npz'), table)
    sx, sy, xfrm = TR.synthetic_scaler_dataset(64, seed=9, samples=20000)
    sx = sx - 90.0
    smodel = TR.ScalerRegressor(noise=0.5)
    TR.Trainer(smodel, torch.nn.functional.mse_loss, device='cuda', batch_size=32, epochs=2,
               validation_split=0.0).fit(sx, sy)
    spath = TR.export_scaler_bundle(smodel.cpu(), str(tmp_path / 'scaler-trained.npz'), xfrm)
    ctx = NativeContext(config_with(demux=dpath, scaler=spath), device_id=0)
    try:
        got = ctx.demux_lstm(x[:256].numpy())
        assert np.abs(got - logp[:256].exp().numpy()).max() < 1e-4
        smodel.eval()
        with torch.no_grad():
            want = smodel(sx[:32]).numpy()
        assert np.abs(ctx.scaler_lstm(sx[:32].numpy()) - want).max() < 2e-4
    finally:
        ctx.close()


def training_windows_of_the_golden_batch(tmp_path):
    """tools/training_windows.py over the golden bundle: the windows of the pushed reads equal the
    ones the REAL reference queued for its classifier (tests/golden/batch0.stages.npz), ids and
    labels line up, and the file feeds the trainer."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import training_windows as TW
    golden = os.path.join(ROOT, 'tests', 'golden')
    bundle = os.path.join(golden, 'batch0.pxr.npz')
    cfg = default_config(inputdir=golden, outputdir=str(tmp_path), read_bundle=bundle, barcoding=True)
    windows, ids, guess, score = TW.collect(cfg, batch_reads=9)          # four batches
    st = np.load(os.path.join(golden, 'batch0.stages.npz'))
    all_ids = [str(r) for r in np.load(bundle)['read_id']]
    pushed = st['pushed'].astype(bool)
    assert ids.tolist() == [r for r, p in zip(all_ids, pushed) if p]
    assert windows.dtype == np.float32 and np.array_equal(windows, st['window'][pushed])
    assert len(guess) == len(score) == len(windows) and ((guess >= -1) & (guess < 4)).all()
    # ... and into the trainer: (windows, labels) -> one epoch
    x = torch.from_numpy(windows)
    y = torch.from_numpy((guess + 1).astype(np.int64))                   # class 0 = decoy
    cost = TR.cost_matrix(5)
    tr = TR.Trainer(TR.DemuxClassifier(), lambda lp, t: TR.weighted_cross_entropy(lp, t, cost), device='cpu',
                    batch_size=8, epochs=1, lr=1e-3, patience=1, output_dir=str(tmp_path / 'fit'))
    hist = tr.fit(x, y)
    assert np.isfinite(hist[-1]['val_loss'])


def test_training_windows_from_a_run(tmp_path, monkeypatch):
    from poreplex_amd import native as N
    from poreplex_amd.worker_persistence import WorkerPersistenceStorage
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle_context import OracleBackedContext
    WorkerPersistenceStorage.reset()
    monkeypatch.setattr(N, 'NativeContext', OracleBackedContext)
    try:
        training_windows_of_the_golden_batch(tmp_path)
    finally:
        WorkerPersistenceStorage.reset()


@pytest.mark.gpu
def test_training_windows_from_a_run_on_the_gpu(tmp_path):
    from poreplex_amd.worker_persistence import WorkerPersistenceStorage
    WorkerPersistenceStorage.reset()
    try:
        training_windows_of_the_golden_batch(tmp_path)
    finally:
        WorkerPersistenceStorage.reset()
