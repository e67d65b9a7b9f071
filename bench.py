#!/usr/bin/env python
"""bench.py -- reads/s of the raw-signal hot path (segment + barcode) on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W.  With N > 1 and no
WORLD_SIZE in the environment this script re-launches itself under torch.distributed.run
(one rank per GPU over RCCL); launched by torch.distributed.run it reads RANK / LOCAL_RANK /
WORLD_SIZE.  A "step" is one pass of the hot path (head pool -> scaler LSTM -> pool + scale +
Viterbi -> barcode window -> demux LSTMs -> result records, + D2H of the records, + the RCCL
label all-gather for N > 1) over one resident batch of synthetic reads; inputs are in HBM
before the timed region.  Rank 0 prints ONE JSON line on stdout.

Workload = BASELINE.json configs[2] (the config the metric "reads/s (segment+barcode)" is
quoted on): a 10 000-read batch of ~60 000-sample reads per GPU, all stages a1-a13.
  --workload segment|polya|chimera|full   configs[1] / configs[3] stage sets
  --scaling strong --total-reads 1000000  configs[4]: ONE seeded run sharded over the ranks
  --base-reads K                          K distinct synthetic reads, tiled on the device into
                                          the resident batch (automatic for big batches:
                                          configs[3] is 12 GB and configs[4] 15 GB of int16
                                          per GPU, never materialised on the host)
"""
import argparse
import copy
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from poreplex_amd import native as N  # noqa: E402
from poreplex_amd.config import default_config  # noqa: E402
from poreplex_amd.distributed import (gather_labels, gather_labels_start,  # noqa: E402
                                      shard_range)
from poreplex_amd.synth import synth_batch  # noqa: E402

# algorithmic work per read (SURVEY.md 8d / DESIGN.md "Roofline accounting")
FLOP_SCALER = 2.0 * 2000 * (48 * 192 + 96 * 192) + 2000 * 192 * 2 + 2 * 48 * 2   # ~111.4 MFLOP
FLOP_BIDIR = 2.0 * 300 * 2 * (48 * 192) + 300 * 2 * 192 * 2
FLOP_TOP = 2.0 * 300 * (160 * 256) + 2 * 64 * 5
PEAK_FP32_MFMA = 157.3e12      # MI355X_MICROARCH.md: dense fp32 MFMA peak, FLOP/s
# dense int8 MFMA peak: 2 x the bf16 rate (MI355X_MICROARCH.md "I8 ~2x bf16 rate (2xK)", bf16 ~2.5 PF dense);
# the guide's own micro-benchmark ceiling for v_mfma_i32_16x16x64_i8 is 3.944 POP/s
PEAK_I8_MFMA = 5.0e15
PEAK_I8_MFMA_MEASURED = 3.944e15     # the guide's measured v_mfma_i32_16x16x64_i8 ceiling (MI355X_MICROARCH.md, matrix-core table)
# PXG_LSTM_Q8 (k_lstm_q8.hip): every float32 multiply-add of a recurrent matrix product becomes EIGHT int8
# digit products (four significance levels); executed MFMAs also multiply the zero bytes that pad a
# 48-unit vector to the instruction's 64-wide k block
Q8_PRODUCTS = 8
OPS_SCALER_Q8 = Q8_PRODUCTS * 2.0 * 2000 * (48 * 192 + 96 * 192)              # algorithmic int8 ops per read
OPS_SCALER_Q8_EXECUTED = 2001 * 4 * 63 * 32768.0 / 16                          # 4 waves x 63 MFMAs per 16 reads and step (round 5: layer 2 unpadded)
PEAK_HBM = 8.0e12              # B/s
# the only throughput the reference publishes (Poreplex 0.1, whole
# pipeline incl. FAST5 I/O, 2x Xeon E5-2687W v3 = 20 cores): 1 339 070 reads in 1 h 37 min
PUBLISHED_READS_PER_S = 1339070 / (97 * 60.0)
TRAFFIC_FILE = os.path.join('profiles', 'r06', 'k_hbm_traffic.json')
# PXG_BENCH_SHARE_GPU=1: every rank of a torchrun launch uses GPU 0 and the collectives go over gloo --
# the real multi-process path (sharding, barriers, max over ranks, label gather, NUMA binding, the
# host-side legs on all ranks at once) with the real kernels on a ONE-GPU box.  A plumbing check:
# the line says so and is not a scaling number.
SHARE_GPU = os.environ.get('PXG_BENCH_SHARE_GPU') == '1'

DTYPE = {'q8': 'i8 x 3 digits, int32 exact sums (LSTM matmuls on the int8 MFMA pipe) + f32 gates / f64 (Viterbi) / i16 in',
         'f32': 'f32 (LSTM MFMA) / f64 (Viterbi) / i16 in'}

STAGES = {
    'demux': (2, 'a1-a13 (scaler LSTM + Viterbi + barcode LSTMs)', 'reads/s (segment+barcode)'),
    'segment': (1, 'a1,a5,a7,a8 (injected scaling)', 'reads/s (normalise+segment)'),
    'polya': (3, 'a1-a17 (+ poly(A) events/DP)', 'reads/s (segment+barcode+polyA)'),
    'chimera': (3, 'a1-a13 + a18/a19 (Guppy block means + window scan)',
                'reads/s (segment+barcode+chimera filter)'),
    'full': (3, 'a1-a19 (+ poly(A) + Guppy block means + window scan)',
             'reads/s (segment+barcode+polyA+chimera filter)'),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--reads', type=int, default=10000, help='reads per GPU per step (weak scaling)')
    ap.add_argument('--samples', type=int, default=60000, help='nominal samples per read')
    ap.add_argument('--workload', choices=sorted(STAGES), default='demux')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak')
    ap.add_argument('--total-reads', type=int, default=1000000,
                    help='--scaling strong: reads of the whole run, sharded over the ranks')
    ap.add_argument('--base-reads', type=int, default=-1,
                    help='distinct reads generated on the host and tiled on the device '
                         '(0 = every read distinct; -1 = 0 up to 16384 reads per GPU, else 2048)')
    ap.add_argument('--cpu-sample', type=int, default=1024,
                    help='reads timed on ONE host core for cpu_baseline (0 = skip)')
    ap.add_argument('--cpu-all-cores-sample', type=int, default=4096,
                    help='reads timed over all physical cores, one process each (0 = skip)')
    ap.add_argument('--seed', type=int, default=924)
    ap.add_argument('--length-dist', choices=['lognormal'], default=None,
                    help='read lengths as a sequencing run produces them (poreplex_amd.synth: median 40 000 samples, 15 %% '
                         'below 30 000, a tail to 1 000 000) instead of --samples +- 10 %%')
    ap.add_argument('--no-run-shaped-leg', action='store_true', help='skip the run-shaped (log-normal lengths) leg of the default line')
    ap.add_argument('--lstm-arith', choices=['q8', 'f32'], default=None,
                    help='arithmetic of the recurrent matmuls (include/pxg.h pxg_lstm_arith); default: the '
                         'config\'s (q8 = exact fixed point on the int8 matrix pipe)')
    ap.add_argument('--no-f32-leg', action='store_true', help='skip the float32-arithmetic comparison leg of the default line')
    ap.add_argument('--no-full-leg', action='store_true', help='skip the configs[3] (poly(A) + chimera filter) leg of the default line')
    ap.add_argument('--no-overlap-test', action='store_true',
                    help='skip the extra PCIe-overlapped steps (profiling runs: keeps the kernel '
                         'statistics to the timed steps)')
    ap.add_argument('--end-to-end', action='store_true',
                    help='files -> labels: every rank writes its shard of the run as a .pxr.npz read '
                         'bundle, then the session driver (poreplex_amd/session.py: loader thread, pinned '
                         'double-buffered batches, facade, sequencing_summary.txt, RCCL all-gather / '
                         'all-reduce) processes it; value = reads / wall of the whole session')
    ap.add_argument('--batch-reads', type=int, default=10000, help='--end-to-end: reads per GPU batch')
    ap.add_argument('--from-fast5', choices=['none', 'gzip', 'vbz'], default=None,
                    help='--end-to-end: the shard is written as multi-read FAST5 files (4 000 reads each, '
                         'Signal uncompressed / gzip / VBZ) and read back by the native reader '
                         '(csrc/pxg_h5.cpp) instead of a read bundle')
    ap.add_argument('--no-fast5-leg', action='store_true', help='skip the FAST5 ingest leg of the default line')
    ap.add_argument('--no-e2e-leg', action='store_true', help='skip the end-to-end session legs of the default line')
    ap.add_argument('--compressed-bundle', action='store_true',
                    help='--end-to-end: the bundle carries encoded samples (pxg_z_*), decoded on the GPU')
    ap.add_argument('--api', choices=['resident', 'process_batch'], default='resident',
                    help="process_batch: value = reads/s through the reference's worker entry point "
                         '(poreplex_amd.signal_analyzer.process_batch(batchid, reads, config) -> list of result '
                         'dicts, pipeline.py:204-205), --in-flight calls kept in flight on one GPU context; the '
                         'default line carries the same figure in extra.process_batch_reads_per_s')
    ap.add_argument('--in-flight', type=int, default=5,
                    help='process_batch leg: worker calls in flight (threads of this process; the '
                         "reference's `parallel`, pipeline.py:96)")
    ap.add_argument('--api-calls', type=int, default=16, help='process_batch leg: timed calls')
    ap.add_argument('--no-api-leg', action='store_true', help='skip the process_batch leg of the default line')
    ap.add_argument('--no-configs4', action='store_true',
                    help='N > 1, weak scaling: skip the extra strong-scaling leg (BASELINE configs[4]: '
                         '--total-reads sharded over the ranks) that the line carries as configs4_strong')
    ap.add_argument('--strong-base-reads', type=int, default=2048,
                    help='distinct reads of the configs4_strong leg (tiled on the device)')
    ap.add_argument('--no-worker-processes-leg', action='store_true',
                    help='API leg: skip the 128-read calls from 8 / 16 worker PROCESSES (the reference\'s ProcessPoolExecutor pattern)')
    ap.add_argument('--no-latency-leg', action='store_true',
                    help='skip the small-batch leg (roofline.latency_form: a 1 024-read batch with and without the latency forms of K2 / K5)')
    return ap.parse_args(argv)


# The CPU rendezvous tests of the multi-rank driver (tests/test_distributed_cpu.py) run this file's main() through
# tests/bench_standin.py, which sets the two names below to a stand-in context class and to its own path.  bench.py
# itself has no way to set them -- no flag, no environment variable: the file that prints the headline can only
# build the HIP context (a line made with a stand-in says data = TEST-STANDIN and value = null).
CONTEXT_CLASS = None
ENTRY_SCRIPT = os.path.abspath(__file__)


def respawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: become N ranks under
    torch.distributed.run (the reference's process-level data parallelism,
    pipeline.py:96,204-205, one worker process per device)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node={}'.format(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), ENTRY_SCRIPT] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def host_description():
    model = 'unknown'
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or os.cpu_count()
    except ImportError:
        physical = os.cpu_count()
    usable = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    # a container may see every core of the host and still be capped by a cgroup CPU quota
    quota = None
    for path, parse_quota in (('/sys/fs/cgroup/cpu.max', lambda t: t.split()),
                              ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', lambda t: [t.strip(), None])):
        try:
            with open(path) as fh:
                q, period = parse_quota(fh.read())
            if period is None:
                with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as fh:
                    period = fh.read().strip()
            if q not in ('max', '-1'):
                quota = float(q) / float(period)
            break
        except (OSError, ValueError):
            continue
    return model, int(physical), int(usable), quota


# ---- CPU baseline over all cores: one process per physical core, read shards ----------
_POOL = {}


def _cpu_worker_init(config, base, mask, use_inject):
    from oracle.pxo import Oracle
    config = copy.deepcopy(config)
    config['signal_processing']['lstm_arith'] = 'f32'      # the CPU's own best arithmetic, see cpu_baseline
    os.environ.pop('PXG_LSTM_ARITH', None)
    _POOL.update(oracle=Oracle(config), base=base, mask=mask, use_inject=use_inject)


def _cpu_worker_run(read_ids):
    b = _POOL['base']
    o = b['offsets']
    parts = [b['arena'][o[i]:o[i + 1]] for i in read_ids]
    arena, off = N.pack_reads(parts)
    inj = b['scale_shift'][read_ids] if _POOL['use_inject'] else None
    t0 = time.perf_counter()
    _POOL['oracle'].process_batch(arena, off, b['calib'][read_ids], inj, _POOL['mask'])
    return time.perf_counter() - t0


def cpu_all_cores(config, base, mask, use_inject, n_sample):
    """The reference's own deployment shape on the CPU: ProcessPoolExecutor(config['parallel'])
    (pipeline.py:96), every worker process running whole reads.  Must run BEFORE the HIP
    runtime is initialised in this process (fork)."""
    import multiprocessing as mp
    model, physical, usable, quota = host_description()
    workers = max(1, min(physical, usable, int(quota) if quota else physical))
    n_base = len(base['offsets']) - 1
    n = max(n_sample, 96 * workers)                        # ~0.7 s of work per core at least
    ids = np.arange(n) % n_base
    shards = [ids[w::workers] for w in range(workers)]
    ctx = mp.get_context('fork')
    with ctx.Pool(workers, initializer=_cpu_worker_init,
                  initargs=(config, base, mask, use_inject)) as pool:
        pool.map(_cpu_worker_run, [s[:2] for s in shards])     # warm: library loaded, pages touched
        t0 = time.perf_counter()
        busy = pool.map(_cpu_worker_run, shards)
        wall = time.perf_counter() - t0
    return {'value': n / wall, 'unit': 'reads/s', 'cores': workers, 'reads': int(n),
            'wall_s': round(wall, 3), 'per_core': n / wall / workers,
            'slowest_worker_s': round(max(busy), 3), 'cgroup_cpu_quota': quota}


def end_to_end(args, config, rank, local_rank, world, dist, standin, base, which, lo, total,
               mask, json_fd):
    """configs[4] as a user gets it: read bundles on disk -> session driver -> labels."""
    import shutil
    import tempfile
    from poreplex_amd.fast5_file import write_bundle
    from poreplex_amd.session import GpuSession
    from poreplex_amd.synth import synth_basecalls
    from poreplex_amd.worker_persistence import WorkerPersistenceStorage
    if standin:
        N.NativeContext = CONTEXT_CLASS
    work = tempfile.mkdtemp(prefix='pxg_e2e_r{}_'.format(rank))
    # the output directory is shared by the ranks (rank 0 stitches their part files)
    outdir = os.path.join(tempfile.gettempdir(), 'pxg_e2e_out_{}_{}'.format(
        os.environ.get('MASTER_PORT', 'solo'), os.getppid() if world > 1 else os.getpid()))
    try:
        o = base['offsets']
        parts = [base['arena'][o[b]:o[b + 1]] for b in which]
        arena, off = N.pack_reads(parts)
        shard = {'offsets': off}
        names = ['shard{:02d}/read{:07d}.fast5'.format(rank, lo + j) for j in range(len(which))]
        ids = ['{:08x}-0000-4000-8000-{:012x}'.format(args.seed, lo + j) for j in range(len(which))]
        t0 = time.perf_counter()
        bcs = synth_basecalls(shard, seed=args.seed + rank)
        if args.from_fast5:
            from poreplex_amd.fast5_write import Fast5Writer
            per_file = 4000
            names = []
            for f0 in range(0, len(which), per_file):
                rel = 'shard{:02d}/reads{:07d}.fast5'.format(rank, lo + f0)
                os.makedirs(os.path.join(work, os.path.dirname(rel)), exist_ok=True)
                with Fast5Writer(os.path.join(work, rel)) as w:
                    for j in range(f0, min(f0 + per_file, len(which))):
                        w.add_read(ids[j], arena[off[j]:off[j + 1]], base['calib'][which[j]], start_time=0,
                                   channel_number='0', run_id='run', sample_id='sample', basecall=bcs[j],
                                   compression=None if args.from_fast5 == 'none' else args.from_fast5)
                        names.append(rel)
            path = None
        else:
            path = os.path.join(work, 'shard.pxr.npz')
            write_bundle(path, arena, off, base['calib'][which], names, ids,
                         basecalls=bcs, compress=args.compressed_bundle)
        t_write = time.perf_counter() - t0
        cfg = default_config(inputdir=work, outputdir=outdir, read_bundle=path,
                             barcoding=True, measure_polya=bool(mask & N.STAGE_POLYA),
                             filter_unsplit_reads=args.workload in ('chimera', 'full'),
                             device_id=local_rank)
        del arena, parts
        t0 = time.perf_counter()
        session = GpuSession(cfg, dist=dist, batch_reads=args.batch_reads)    # context + bundle load
        t_open = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        out = session.run(list(zip(names, ids)), presharded_at=lo)
        wall = time.perf_counter() - t0
        t0 = time.perf_counter()
        session.close()                   # the staging arenas' page locks come off here (reported, not in `value`)
        t_close = time.perf_counter() - t0
        n_ranks = 1
        if dist is not None:
            import torch
            t = torch.tensor([wall, 1.0], dtype=torch.float64, device=collective_device(standin))
            dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
            dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
            wall, n_ranks = float(t[0].item()), int(round(t[1].item()))
        if rank == 0:
            info = session.ctx.device_info()
            counts = out['counts']
            from poreplex_amd.distributed import LABEL_NAMES
            with open(os.path.join(cfg['outputdir'], 'sequencing_summary.txt')) as fh:
                n_rows = sum(1 for _ in fh) - 1
            line = {
                'metric': 'reads/s (end-to-end demux: {} on disk -> barcode labels + '
                          'sequencing_summary.txt)'.format('FAST5 files' if args.from_fast5 else 'read bundles'),
                'value': None if standin else total / wall, 'unit': 'reads/s', 'n_gpus': world,
                'steps': out['batches'], 'warmup': 0, 'ms_per_step': wall / max(out['batches'], 1) * 1e3,
                'higher_is_better': True, 'scaling': args.scaling,
                'vs_baseline': None,
                'vs_baseline_basis': 'no published figure for this metric; extra.vs_published_whole_pipeline = value / 230 '
                                     'reads/s (BASELINE.md section 1: Poreplex 0.1 whole pipeline, pre-basecalled FAST5 -> '
                                     'FASTQ, 20 Xeon cores -- the closest published counterpart of this end-to-end figure)',
                'dtype': 'f32 (LSTM MFMA) / f64 (Viterbi) / i16 in',
                'data': 'TEST-STANDIN (no GPU work timed)' if standin else ('synthetic (ranks SHARING one GPU: a plumbing check, not a scaling number)' if SHARE_GPU else 'synthetic'),
                'config': {'workload': 'BASELINE configs[4] shape: {} reads over {} GPU(s) x ~{} int16 samples, '
                                       'files -> session driver -> sinks, stages {}'.format(
                                           total, world, args.samples, STAGES[args.workload][1]),
                           'reads_per_gpu': [out['reads_this_rank']], 'batch_reads': args.batch_reads,
                           'samples_per_read': args.samples, 'device': info['name'], 'arch': info['arch']},
                'roofline': None, 'cpu_baseline': None, 'concordance': None,
                'extra': {'vs_published_whole_pipeline': None if standin else total / wall / PUBLISHED_READS_PER_S,
                          'session_timing_rank0': {k: (round(v, 4) if isinstance(v, float) else v) for k, v in out['timing'].items()},
                          'bundle_write_s': round(t_write, 3), 'compressed_bundle': bool(args.compressed_bundle), 'from_fast5': args.from_fast5, 'context_and_bundle_open_s': round(t_open, 3), 'session_close_s': round(t_close, 3),
                          'reads_labelled_pass': int(counts[LABEL_NAMES.index('pass')].sum()),
                          'reads_with_barcode': int(counts[:, 1:].sum()), 'summary_rows': n_rows,
                          'labels_gathered': int(len(out['labels'])),
                          'labels_read_index_unique': bool(len(np.unique(out['labels']['read_index'])) == len(out['labels'])),
                          'ranks_counted_by_collective': n_ranks,
                          'host_us_per_read_rank0': round((out['timing']['facade_s'] + out['timing']['sink_s'])
                                                          / max(out['reads_this_rank'], 1) * 1e6, 2)},
            }
            os.write(json_fd, (json.dumps(line) + '\n').encode())
        WorkerPersistenceStorage.reset()
    finally:
        shutil.rmtree(work, ignore_errors=True)
        if rank == 0:
            shutil.rmtree(outdir, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


def process_batch_leg(args, base, which, lo, mask, local_rank, compressed, resident_records=None):
    """The drop-in API north_star names, as the reference's pipeline submits it
    (pipeline.py:193-229: `parallel` process_batch(batchid, reads, config) calls in flight, each
    returning the list of result dicts): the reads sit in a read bundle on disk, every call
    copies its samples to the GPU again (page-locked bundle arena -> spare input slot on the
    copy stream), runs every stage and builds the dicts.  Nothing stays resident between calls.
    Returns rates for one call at a time and for `--in-flight` calls overlapping on one context."""
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from poreplex_amd.fast5_file import write_bundle
    from poreplex_amd.signal_analyzer import process_batch
    from poreplex_amd.synth import synth_basecalls
    from poreplex_amd.worker_persistence import WorkerPersistenceStorage
    work = tempfile.mkdtemp(prefix='pxg_api_')
    try:
        o = base['offsets']
        if len(which) == len(o) - 1 and np.array_equal(which, np.arange(len(which))):
            arena, off = base['arena'], o
        else:
            arena, off = N.pack_reads([base['arena'][o[b]:o[b + 1]] for b in which])
        n = len(which)
        names = ['api/read{:07d}.fast5'.format(lo + j) for j in range(n)]
        ids = ['{:08x}-0000-4000-8000-{:012x}'.format(args.seed, lo + j) for j in range(n)]
        path = os.path.join(work, 'api.pxr.npz')
        t0 = time.perf_counter()
        write_bundle(path, arena, off, base['calib'][which], names, ids,
                     basecalls=synth_basecalls({'offsets': off}, seed=args.seed), compress=compressed)
        t_write = time.perf_counter() - t0
        cfg = default_config(inputdir=work, outputdir=work, read_bundle=path, barcoding=bool(mask & N.STAGE_BARCODE),
                             measure_polya=bool(mask & N.STAGE_POLYA),
                             filter_unsplit_reads=args.workload in ('chimera', 'full'), device_id=local_rank)
        reads = list(zip(names, ids))
        WorkerPersistenceStorage.reset()
        t0 = time.perf_counter()
        first = process_batch(0, reads, cfg)          # context, bundle load, page-locking: once per worker
        t_first = time.perf_counter() - t0
        if isinstance(first, tuple):
            raise N.PxgError('process_batch failed: {}'.format(first[1]))
        t0 = time.perf_counter()
        for k in range(2):
            process_batch(1 + k, reads, cfg)
        serial = 2 * n / (time.perf_counter() - t0)
        calls = max(args.api_calls, 1)
        from poreplex_amd import signal_analyzer as SA
        SA.CALL_TRACE = trace = []
        plain_before = getattr(SA, 'PLAIN_RUN_CALLS', 0)
        # results are checked and DROPPED as they arrive, like a pipeline that hands them to its sinks
        # (holding on to millions of dicts is not the API's cost: it makes the cyclic collector walk
        # them all, under the GIL every worker thread needs)
        verdict = {'bad': None, 'same': True, 'last': None}

        def one_call(k):
            r = process_batch(10 + k, reads, cfg)
            if isinstance(r, tuple):
                verdict['bad'] = r
            else:
                verdict['same'] = verdict['same'] and len(r) == len(first) and r[0] == first[0] and r[-1] == first[-1]
                if k == calls - 1:
                    verdict['same'] = verdict['same'] and r == first
                    verdict['last'] = r
            return k
        if os.environ.get('PXG_BENCH_SWITCH_INTERVAL'):       # (experiment knob: the interpreter's GIL hand-over interval)
            sys.setswitchinterval(float(os.environ['PXG_BENCH_SWITCH_INTERVAL']))
        if os.environ.get('PXG_BENCH_GC_OFF'):                 # (experiment knob: no cyclic collections under the worker threads)
            import gc
            gc.collect()
            gc.freeze()
            gc.disable()
        with ThreadPoolExecutor(max(args.in_flight, 1)) as pool:
            t0 = time.perf_counter()
            list(pool.map(one_call, range(calls)))
            wall = time.perf_counter() - t0
        SA.CALL_TRACE = None
        tr = np.array(trace)
        phases = {'prepare_ms': float((tr[:, 1] - tr[:, 0]).mean() * 1e3),
                  'gpu_pass_ms': float((tr[:, 2] - tr[:, 1]).mean() * 1e3),
                  'result_dicts_ms': float((tr[:, 3] - tr[:, 2]).mean() * 1e3),
                  'gpu_pass_done_spacing_ms': float(np.diff(np.sort(tr[:, 2])).mean() * 1e3) if len(tr) > 1 else None}
        if verdict['bad'] is not None:
            raise N.PxgError('process_batch failed: {}'.format(verdict['bad'][1]))
        last = verdict['last']
        out = {'reads_per_s': calls * n / wall, 'one_call_at_a_time_reads_per_s': serial,
               'calls': calls, 'in_flight': args.in_flight, 'reads_per_call': n, 'ms_per_call': wall / calls * 1e3,
               'first_call_s': round(t_first, 3), 'bundle_write_s': round(t_write, 3),
               'compressed_bundle': bool(compressed), 'dicts_returned': len(last),
               'dict_builder': 'csrc/_pxgpy' if N.load_pyhost() is not None else 'python loop',
               'results_identical_across_calls': bool(verdict['same']),
               # calls judged and reported in one C pass, without a batch table (SignalAnalyzer.process_plain_run)
               'calls_on_the_plain_run_path': getattr(SA, 'PLAIN_RUN_CALLS', 0) - plain_before,
               'mean_phase_ms_per_call': {k: (round(v, 2) if v is not None else None) for k, v in phases.items()}}
        try:                               # small calls that met in the pipeline ran as one batch (include/pxg.h)
            probe = SA.SignalAnalyzer(cfg, 0)
            groups, merged = probe.ctx.merge_stats()
            probe.close()
            out['merge_stats'] = {'batches': groups, 'calls_they_carried': merged}
        except Exception:
            pass
        if resident_records is not None and len(resident_records) == n:
            # the dicts against the records of the resident loop (same reads, same stages)
            st = [N.STATUS_NAMES[c] for c in resident_records['status'].tolist()]
            called = resident_records['bc_called'].tolist()
            label = resident_records['bc_label'].tolist()
            by_id = {r.get('read_id'): r for r in last}
            diff = 0
            for j, rid in enumerate(ids):
                r = by_id[rid]
                # a status the GPU pass decided must be the dict's; a GPU 'okay' may still fail later
                # rules (adapter / basecall / length), but never with a GPU-side status
                if st[j] != 'okay':
                    diff += int(r['status'] != st[j])
                else:
                    diff += int(r['status'] in ('scaler_signal_too_short', 'scaling_qc_fail'))
                diff += int(r.get('barcode') != (label[j] if called[j] else None))
            out['barcode_or_status_mismatch_vs_resident_records'] = diff
        WorkerPersistenceStorage.reset()
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def fast5_calls_leg(args, base, which, lo, mask, local_rank, resident_records, n=128, file_reads=1024, threads=32,
                    calls=256):
    """The reference's batch size over the reference's INPUT: `file_reads` reads of the batch in one multi-read FAST5
    file (no read bundle), served as consecutive `n`-read process_batch calls in the file's read order -- one at a time
    and from `threads` worker threads that share the context.  Every call opens its reads through the native reader
    (a per-call bundle: SignalLoader.fast5_run_bundle), copies the samples to the GPU, runs every stage and builds the
    dicts; the dicts of the first pass are checked against the records of the resident loop."""
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from poreplex_amd import signal_analyzer as SA
    from poreplex_amd.fast5_file import get_read_ids
    from poreplex_amd.fast5_write import Fast5Writer
    from poreplex_amd.synth import synth_basecalls
    from poreplex_amd.worker_persistence import WorkerPersistenceStorage
    work = tempfile.mkdtemp(prefix='pxg_api_f5_')
    try:
        total = min(file_reads, len(which)) // n * n
        if total < n:
            raise N.PxgError('fewer than {} reads in the batch'.format(n))
        o = base['offsets']
        raws = [base['arena'][o[b]:o[b + 1]] for b in which[:total]]
        bcs = synth_basecalls({'offsets': np.concatenate([[0], np.cumsum([len(r) for r in raws])])}, seed=args.seed)
        ids = ['{:08x}-0000-4000-8000-{:012x}'.format(args.seed, lo + j) for j in range(total)]
        t0 = time.perf_counter()
        with Fast5Writer(os.path.join(work, 'run.fast5')) as w:
            for j in range(total):
                w.add_read(ids[j], raws[j], base['calib'][which[j]], start_time=j, channel_number=str(1 + j % 512),
                           basecall=bcs[j])
        t_write = time.perf_counter() - t0
        cfg = default_config(inputdir=work, outputdir=work, barcoding=bool(mask & N.STAGE_BARCODE),
                             measure_polya=bool(mask & N.STAGE_POLYA),
                             filter_unsplit_reads=args.workload in ('chimera', 'full'), device_id=local_rank)
        keys = get_read_ids('run.fast5', work)
        slices = [keys[k:k + n] for k in range(0, total, n)]
        WorkerPersistenceStorage.reset()
        first = [SA.process_batch(k, reads, cfg) for k, reads in enumerate(slices)]     # context + file open: once per worker
        bad = [r for r in first if isinstance(r, tuple)]
        if bad:
            raise N.PxgError('process_batch failed: {}'.format(bad[0][1]))
        plain_before = SA.PLAIN_RUN_CALLS
        t0 = time.perf_counter()
        for k, reads in enumerate(slices):
            SA.process_batch(100 + k, reads, cfg)
        serial = total / (time.perf_counter() - t0)
        failed = []

        def one_call(k):
            r = SA.process_batch(200 + k, slices[k % len(slices)], cfg)
            if isinstance(r, tuple) or len(r) != n:
                failed.append(r)
            return k
        with ThreadPoolExecutor(threads) as pool:
            t0 = time.perf_counter()
            list(pool.map(one_call, range(calls)))
            wall = time.perf_counter() - t0
        if failed:
            raise N.PxgError('process_batch failed: {}'.format(failed[0][1] if isinstance(failed[0], tuple) else 'short result'))
        out = {'reads_per_s': calls * n / wall, 'one_call_at_a_time_reads_per_s': serial, 'calls': calls, 'threads': threads,
               'reads_per_call': n, 'reads_in_the_file': total, 'file_MB': round(os.path.getsize(os.path.join(work, 'run.fast5')) / 1e6, 1),
               'file_write_s': round(t_write, 2), 'compression': 'none',
               'calls_on_the_plain_run_path': SA.PLAIN_RUN_CALLS - plain_before, 'timed_calls': len(slices) + calls}
        if resident_records is not None and len(resident_records) >= total:
            at = {r: j for j, r in enumerate(ids)}
            st = [N.STATUS_NAMES[c] for c in resident_records['status'][:total].tolist()]
            called, label = resident_records['bc_called'][:total].tolist(), resident_records['bc_label'][:total].tolist()
            diff = 0
            for part in first:
                for r in part:
                    j = at[r['read_id']]
                    if st[j] != 'okay':
                        diff += int(r['status'] != st[j])
                    else:
                        diff += int(r['status'] in ('scaler_signal_too_short', 'scaling_qc_fail'))
                    diff += int(r.get('barcode') != (label[j] if called[j] else None))
            out['barcode_or_status_mismatch_vs_resident_records'] = diff
        WorkerPersistenceStorage.reset()
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def worker_processes_leg(args, base, which, lo, mask, local_rank, workers=(8, 16), calls_per_worker=40, n=128):
    """The reference's OWN call pattern for its per-read processor (pipeline.py:96,193-205): a ProcessPoolExecutor of
    `parallel` worker processes, each running process_batch(batchid, reads, config) on 128-read batches
    (commandline.py:402) and handing the list of result dicts back through the executor's pipe.  Every worker process
    has its own interpreter and its own context on the ONE GPU (WorkerPersistenceStorage in that process); a
    128-read batch takes the latency forms of the LSTM kernels on 32 of the 256 CUs, so the workers' kernels run side by
    side.  Every worker builds its context in the pool's initializer and 20 x workers warm-up calls are served before the
    clock starts; timed: `calls_per_worker` x workers calls submitted at once, the result lists arriving in the parent
    (pickling included)."""
    import multiprocessing as mp
    import shutil
    import tempfile
    from concurrent.futures import ProcessPoolExecutor
    from poreplex_amd.fast5_file import write_bundle
    from poreplex_amd.signal_analyzer import process_batch       # (the pool pickles it by name: the workers import the package)
    from poreplex_amd.synth import synth_basecalls
    from poreplex_amd.worker_persistence import warm_up
    work = tempfile.mkdtemp(prefix='pxg_wp_')
    out = {'reads_per_call': n, 'calls_per_worker': calls_per_worker,
           'pattern': 'ProcessPoolExecutor(workers).submit(process_batch, batchid, reads, config): pipeline.py:96,204-205'}
    try:
        o = base['offsets']
        arena, off = N.pack_reads([base['arena'][o[b]:o[b + 1]] for b in which[:n]])
        names = ['wp/read{:07d}.fast5'.format(lo + j) for j in range(n)]
        ids = ['{:08x}-0000-4000-8000-{:012x}'.format(args.seed, lo + j) for j in range(n)]
        path = os.path.join(work, 'wp.pxr.npz')
        write_bundle(path, arena, off, base['calib'][which[:n]], names, ids,
                     basecalls=synth_basecalls({'offsets': off}, seed=args.seed), compress=False)
        cfg = default_config(inputdir=work, outputdir=work, read_bundle=path, barcoding=bool(mask & N.STAGE_BARCODE),
                             measure_polya=bool(mask & N.STAGE_POLYA), filter_unsplit_reads=False, device_id=local_rank)
        reads = list(zip(names, ids))
        for w in workers:
            t_start = time.perf_counter()
            with ProcessPoolExecutor(w, mp_context=mp.get_context('spawn'), initializer=warm_up, initargs=(cfg,)) as pool:
                warm = [pool.submit(process_batch, 1000000 + k, reads, cfg) for k in range(20 * w)]
                first = [f.result(timeout=300) for f in warm][0]
                if isinstance(first, tuple):
                    raise N.PxgError('process_batch failed in a worker process: {}'.format(first[1]))
                t_up = time.perf_counter() - t_start
                t0 = time.perf_counter()
                futs = [pool.submit(process_batch, k, reads, cfg) for k in range(calls_per_worker * w)]
                got = [f.result(timeout=300) for f in futs]
                wall = time.perf_counter() - t0
            same = all(isinstance(g, list) and len(g) == len(first) and g[0] == first[0] and g[-1] == first[-1] for g in got)
            out['{}_workers'.format(w)] = {'reads_per_s': len(futs) * n / wall, 'calls': len(futs), 'ms_per_call': wall / len(futs) * 1e3,
                                          'pool_start_and_warm_up_s': round(t_up, 2), 'dicts_per_call': len(first),
                                          'results_identical_across_calls': bool(same)}
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def fast5_ingest_leg(args, base, which, n_reads=2048):
    """FAST5 -> staging arena, host only (never `value`): `n_reads` of the batch written as
    multi-read FAST5 files with poreplex_amd/fast5_write.py, then opened and decoded into one
    arena by the native reader (csrc/pxg_h5.cpp: memory map, HDF5 structures parsed in C++, gzip /
    zstd + streamvbyte undone on host threads) -- the rate the session's loader thread can feed
    the GPU at from real files, per compression."""
    import shutil
    import tempfile
    from poreplex_amd import fast5_file as F5
    from poreplex_amd.fast5_write import Fast5Writer
    from poreplex_amd.synth import synth_basecalls
    out = {'reads': 0, 'threads': F5.host_threads()}
    work = tempfile.mkdtemp(prefix='pxg_f5_')
    try:
        n = min(n_reads, len(which))
        o = base['offsets']
        raws = [base['arena'][o[b]:o[b + 1]] for b in which[:n]]
        bcs = synth_basecalls({'offsets': np.concatenate([[0], np.cumsum([len(r) for r in raws])])}, seed=args.seed)
        out['reads'] = n
        stage = np.empty(sum(len(r) for r in raws) + 16, dtype=np.int16)
        stage.fill(0)                  # a staging arena that has been used before, like the session's from its third batch on
        for mode, count in (('none', n), ('vbz', n), ('gzip', min(n, 512))):
            path = os.path.join(work, mode + '.fast5')
            t0 = time.perf_counter()
            try:
                with Fast5Writer(path) as w:
                    for j in range(count):
                        w.add_read('{:08x}-0000-4000-8000-{:012x}'.format(args.seed, j), raws[j], base['calib'][which[j]],
                                   start_time=j, channel_number=str(1 + j % 512), basecall=bcs[j],
                                   compression=None if mode == 'none' else mode)
            except OSError as exc:                       # no libzstd on this host: VBZ cannot be written
                out[mode] = {'error': str(exc)}
                continue
            t_write = time.perf_counter() - t0
            before = dict(F5.TIMING)
            t0 = time.perf_counter()
            f = F5.Fast5File(path)
            bundle = F5.Fast5Batch.from_runs([(f, mode + '.fast5', 0, f.n)]).as_bundle(reserve=lambda k: stage[:k])
            dt = time.perf_counter() - t0
            assert not bundle.signal_status.any() and np.array_equal(bundle.samples(count - 1), raws[count - 1])
            out[mode] = {'reads_per_s': count / dt, 'reads': count, 'file_MB': round(os.path.getsize(path) / 1e6, 1),
                         'samples_GBps': float(bundle.d['offsets'][-1]) * 2 / dt / 1e9, 'write_s': round(t_write, 2),
                         # where the loader's time goes: group walk + metadata, copy / decode of the samples, basecall text
                         'ms': {'total': round(dt * 1e3, 2),
                                'walk': round((F5.TIMING['walk_s'] - before['walk_s']) * 1e3, 2),
                                'signals': round((F5.TIMING['signals_s'] - before['signals_s']) * 1e3, 2),
                                'text': round((F5.TIMING['text_s'] - before['text_s']) * 1e3, 2)}}
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def end_to_end_legs(args):
    """The session driver from files, as part of the DEFAULT line (so that the driver's own run
    carries it): `bench.py --end-to-end` in a child process, once from an encoded read bundle and once
    from uncompressed multi-read FAST5 (12 and 20 batches: fill and drain included -- the first two batches of a
    session load into staging arenas nobody has touched yet, the last one drains alone; `loader_ms_per_batch` is
    the median of the loader thread's time per batch from the third batch on; longer runs are in profiles/).
    Errors are reported, never hidden."""
    import subprocess
    out = {'batch_reads': 10000}
    for name, reads, flags in (('encoded_bundle', 120000, ['--compressed-bundle']),
                               ('fast5_uncompressed', 200000, ['--from-fast5', 'none'])):
        cmd = [sys.executable, os.path.abspath(__file__), '--end-to-end', '--reads', str(reads), '--batch-reads', '10000',
               '--samples', str(args.samples), '--seed', str(args.seed), '--cpu-sample', '0', '--cpu-all-cores-sample', '0'] + flags
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            line = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
            timing = line['extra']['session_timing_rank0']
            steady = sorted(timing.get('load_ms', [])[2:])
            out[name] = {'reads_per_s': line['value'], 'reads': reads, 'batches': line['steps'],
                         'loader_s': timing['load_s'],
                         'loader_ms_per_batch': steady[len(steady) // 2] if steady else None,
                         'summary_rows': line['extra']['summary_rows']}
        except Exception as exc:                       # noqa: BLE001
            out[name] = {'error': '{}: {}'.format(type(exc).__name__, exc)}
    return out


def collective_device(standin):
    """Where the tensors of the bench's own collectives live: the GPU under RCCL, the host under
    gloo (the CPU test stand-in, and PXG_BENCH_SHARE_GPU=1 -- several ranks on ONE GPU)."""
    return 'cpu' if standin or SHARE_GPU else 'cuda'


def rank_stats(value, dist, standin):
    """(min, mean, max) of one number over the ranks (RCCL / gloo all-reduce); None stays None."""
    if dist is None:
        return {'min': value, 'mean': value, 'max': value}
    import torch
    dev = collective_device(standin)
    has = torch.tensor([0.0 if value is None else 1.0], dtype=torch.float64, device=dev)
    dist.all_reduce(has, op=dist.ReduceOp.MIN)
    v = float(value) if value is not None else 0.0
    lo, hi, tot = (torch.tensor([v], dtype=torch.float64, device=dev) for _ in range(3))
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    if float(has.item()) == 0.0:
        return None
    return {'min': float(lo.item()), 'mean': float(tot.item()) / dist.get_world_size(), 'max': float(hi.item())}


def pcie_legs(ctx, base, inject, step, n_local, args, kernel_rate):
    """PCIe-inclusive rates of THIS rank (never `value`): the double-buffered loader path with raw
    int16 samples and with encoded samples, and the H2D ceiling of the link.  At N > 1 every rank
    runs it at the same time: eight ranks staging ~57 GB/s each is the host-side load that can
    fail to scale (host DRAM, NUMA, PCIe roots) while the kernels scale trivially."""
    out = {}
    nbytes_in = base['arena'].nbytes
    # what gets page-locked is an array on mapped pages of its own (native.pinnable: a copy when the array may live in
    # the malloc heap -- profiles/r05/fault_hunt.md)
    arena = N.pinnable(base['arena'])
    ctx.pin(arena)
    n_over = min(args.steps, 5)
    ctx.sync()
    o0 = time.perf_counter()
    for _ in range(n_over):
        step()
        ctx.stage(arena, base['offsets'], base['calib'], inject)
        ctx.download()
        ctx.swap()
    ctx.sync()
    o_s = (time.perf_counter() - o0) / n_over
    out['pcie_overlapped_reads_per_s'] = n_local / o_s
    h0 = time.perf_counter()
    for _ in range(2):
        ctx.stage(arena, base['offsets'], base['calib'], inject)
        ctx.swap()                      # waits for the staged copies
    h_s = (time.perf_counter() - h0) / 2
    out['h2d_GBps'] = nbytes_in / h_s / 1e9
    out['pcie_bound_reads_per_s'] = n_local / h_s
    out['pcie_overlap_efficiency'] = (n_local / o_s) / min(n_local / h_s, kernel_rate)
    ctx.unpin(arena)
    del arena
    z, chunks, _ = N.z_encode(base['arena'], base['offsets'])        # offline step, not timed
    z, chunks = N.pinnable(z), N.pinnable(chunks)
    enc = N.EncodedSamples(z, chunks, 0, 0, len(base['arena']))
    ctx.pin(z)
    ctx.pin(chunks)
    ctx.sync()
    z0 = time.perf_counter()
    for _ in range(n_over):
        step()
        ctx.stage_z(enc, base['offsets'], base['calib'], inject)
        ctx.download()
        ctx.swap()
    ctx.sync()
    z_s = (time.perf_counter() - z0) / n_over
    out['pcie_overlapped_encoded_reads_per_s'] = n_local / z_s
    out['encoded_bytes_per_sample'] = (z.nbytes + chunks.nbytes) / max(len(base['arena']), 1)
    ctx.unpin(z)
    ctx.unpin(chunks)
    return out


def strong_leg(args, ctx, dist, rank, world, mask, standin, force_dist, barrier):
    """BASELINE configs[4] beside the weak value: ONE seeded run of --total-reads reads, global
    read i = distinct read i mod K, rank r owns shard_range(total, r, N) and tiles its shard on
    the device; every step gathers the label records of the whole run."""
    total = args.total_reads
    if SHARE_GPU:                      # the ranks' shards share ONE GPU's HBM here: a quarter of the run
        total = min(total, 250000)
    lo, hi = shard_range(total, rank, world)
    n_local = hi - lo
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    K = max(1, min(args.strong_base_reads, total))
    base = synth_batch(K, seed=args.seed, samples_per_read=args.samples)      # the same on every rank
    res_buf = N.page_exclusive(n_local, N.RESULT_DTYPE, fill=0)
    failure = None
    try:                               # a rank that cannot hold its shard must not leave the others in a collective
        ctx.upload_tiled(n_local, base['arena'], base['offsets'], base['calib'], None, phase=lo % K)
        if not standin:
            ctx.pin(res_buf)
        ctx.run(mask)
        res = ctx.download(res_buf)
    except Exception as exc:
        failure = '{}: {}'.format(type(exc).__name__, exc)
    if dist is not None:
        import torch
        ok = torch.tensor([0 if failure else 1], dtype=torch.int32, device=collective_device(standin))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and failure is None:
            failure = 'another rank could not hold its shard'
    if failure:
        return {'value': None, 'error': failure, 'total_reads': total, 'reads_per_gpu': sizes}
    labels = gather_labels(res, dist, first_index=lo, sizes=sizes, force=force_dist)
    barrier()
    steps = max(2, min(args.steps, 3))
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.run(mask)
        res = ctx.download(res_buf)
        labels = gather_labels(res, dist, first_index=lo, sizes=sizes, force=force_dist)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=collective_device(standin))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not standin:
        ctx.unpin(res_buf)
    return {'value': None if standin else total * steps / elapsed, 'unit': 'reads/s', 'scaling': 'strong',
            'total_reads': total, 'reads_per_gpu': sizes, 'steps': steps, 'ms_per_step': elapsed / steps * 1e3,
            'distinct_reads': K, 'labels_gathered': int(len(labels)),
            'labels_read_index_unique': bool(len(np.unique(labels['read_index'])) == len(labels) == total),
            'reads_ok_this_rank': int((res['status'] == 0).sum())}


FLIPS_FILE = os.path.join('profiles', 'r04', 'decision_flips.json')
FLIPS_GPU_FILE = os.path.join('profiles', 'r05', 'decision_flips_gpu_160k_reads.json')
BOUNDS_FILE = os.path.join('profiles', 'r06', 'full_kernel_bounds.json')
BOUNDS_DEMUX_FILE = os.path.join('profiles', 'r06', 'demux_kernel_bounds.json')      # PMC passes of the default workload
ROCPROF_STATS = {'demux': os.path.join('profiles', 'r06', 'k_demux_kernel_stats.csv'),
                 'full': os.path.join('profiles', 'r06', 'k_full_kernel_stats.csv')}


def unpinned_rows_block():
    """Rows whose third-party numerics (TensorFlow LSTMs: a4, a12; pomegranate Viterbi: a7) cannot
    run in this image: "GPU == oracle" above proves the kernels, not the restatement.  What a
    different-but-correct implementation could move is MEASURED -- tools/decision_flips_gpu.py (both product
    arithmetics and the exact float64 Keras equations on >= 20 000 bench reads, an adversarial read set and
    two sets built ON the scaling-QC bounds / the calling threshold) and tests/test_decision_flips.py (every
    formula variant of the pomegranate Viterbi, bench + adversarial reads); the committed numbers of those
    measurements ride along here -- static, from the files named, not re-measured by this run."""
    out = {'unpinned_rows': ['a4', 'a7', 'a12']}
    block = {}
    try:
        with open(os.path.join(ROOT, FLIPS_GPU_FILE)) as fh:
            d = json.load(fh)
        keep = ('reads', 'scaling_qc_flips', 'status_flips', 'reads_with_a_segment_boundary_moved', 'adapter_found_flips',
                'window_gate_flips', 'windows_compared', 'argmax_flips', 'called_uncalled_flips',
                'called_uncalled_flips_at_threshold', 'barcodes_called', 'scaler_pred_max_abs_diff',
                'softmax_max_abs_diff_whole_pipeline', 'softmax_max_abs_diff')
        sets = {}
        for st in d['sets']:
            if 'q8_vs_f32' in st:
                sets[st['set']] = {pair: {k: v for k, v in st[pair].items() if k in keep}
                                   for pair in ('q8_vs_f32', 'q8_vs_f64', 'f32_vs_f64')}
            else:
                sets[st['set']] = {k: v for k, v in st.items() if k != 'set'}
        block['a4_a12_q8_f32_and_exact_float64_networks'] = dict(
            sets, source='static: {} (python tools/decision_flips_gpu.py, MI355X)'.format(FLIPS_GPU_FILE),
            meaning='q8 / f32 = the two product arithmetics, f64 = exact Keras equations in float64 around the same '
                    'pooling / Viterbi / window stages; qc-edge and call-edge are sets built within 1e-4 of a scaling-QC '
                    'bound and within 1e-3 of the calling threshold')
    except (OSError, KeyError, ValueError):
        pass
    try:
        with open(os.path.join(ROOT, FLIPS_FILE)) as fh:
            d = json.load(fh)
        for key, name in (('viterbi_side', 'a7_viterbi_formula_variants'),
                          ('viterbi_side_adversarial', 'a7_viterbi_formula_variants_adversarial_reads')):
            vs = d.get(key)
            if not vs:
                continue
            worst = {k: max(v[k] for v in vs['variants']) for k in (
                'reads_with_a_segment_boundary_moved', 'adapter_found_flips',
                'bench_reads_candidate_list_changed', 'chimera_reads_candidate_list_changed')}
            block[name] = dict(
                worst, variants=len(vs['variants']), reads_segmented=vs['reads_segmented'],
                scan_windows=vs['scan_windows'], chimera_reads=vs['reads_scanned']['chimera'],
                source='static: {} (python tests/test_decision_flips.py)'.format(FLIPS_FILE),
                meaning='worst count over the variants (mixture as logsumexp / log-sum of pdfs, textbook '
                        'log-pdf, bake() renormalisation, preset in-edge order, all at once, float64 and '
                        'longdouble)')
    except (OSError, KeyError, ValueError):
        pass
    out['unpinned_decision_flips'] = block or None
    return out


def run_shaped_leg(args, ctx, mask, n_reads, distinct=1024, config=None, local_rank=0):
    """The same stages over reads with a sequencing run's length distribution (synth.py length_dist=
    'lognormal'): reads/s, samples/s and the per-stage times, measured like the headline (resident batch,
    kernels + D2H of the records) -- and once more on a context with the two exact savings such a run exposes
    switched off (reads in length order, K2 behind the shared zero-pad prefix: DESIGN 3, round 4), records
    compared."""
    rb = synth_batch(distinct, seed=args.seed + 77, length_dist='lognormal')
    lens = np.diff(rb['offsets'])

    def measure(c):
        c.upload_tiled(n_reads, rb['arena'], rb['offsets'], rb['calib'], None, phase=0)
        buf = np.zeros(n_reads, dtype=N.RESULT_DTYPE)
        for _ in range(2):
            c.run(mask)
            res = c.download(buf)
        c.sync()
        steps = max(3, min(args.steps, 8))
        acc = {k: 0.0 for k in N.TIMER_NAMES}
        t0 = time.perf_counter()
        for _ in range(steps):
            c.run(mask)
            res = c.download(buf)
            times, _ = c.stage_times()
            for k in acc:
                acc[k] += times[k]
        c.sync()
        return time.perf_counter() - t0, steps, acc, res.copy()

    wall, steps, acc, res = measure(ctx)
    plain = None
    if config is not None:
        os.environ['PXG_NO_LENGTH_ORDER'] = os.environ['PXG_NO_PREFIX_SKIP'] = '1'
        try:
            c2 = N.NativeContext(config, device_id=local_rank)
        finally:
            del os.environ['PXG_NO_LENGTH_ORDER'], os.environ['PXG_NO_PREFIX_SKIP']
        try:
            w2, s2, a2, r2 = measure(c2)
            plain = {'reads_per_s': n_reads * s2 / w2, 'ms_per_step': w2 / s2 * 1e3,
                     'stage_ms': {k: round(v / s2, 4) for k, v in a2.items() if k in ('scaler_lstm', 'segment', 'total')},
                     'records_identical': bool(res.tobytes() == r2.tobytes())}
        finally:
            c2.close()
    tiled = lens[np.arange(n_reads) % distinct]
    return {
        'reads_per_s': n_reads * steps / wall, 'ms_per_step': wall / steps * 1e3, 'steps': steps,
        'samples_per_s': float(tiled.sum()) * steps / wall,
        'stage_ms': {k: round(v / steps, 4) for k, v in acc.items()},
        'without_length_order_and_prefix_skip': plain,
        'reads': n_reads, 'distinct_reads': distinct, 'tiled_on_device': True,
        'length_samples': {'min': int(lens.min()), 'median': float(np.median(lens)), 'mean': float(lens.mean()),
                           'max': int(lens.max()), 'below_30000': float((lens < 30000).mean()),
                           'above_100000': float((lens > 100000).mean())},
        'statuses': {N.STATUS_NAMES[int(k)]: int(v) for k, v in zip(*np.unique(res['status'], return_counts=True))},
    }


def f32_leg(args, config, local_rank, base, inject, mask, res_q8, stage_ms_q8):
    """A second context in PXG_LSTM_F32 on the same resident batch: steps timed the same way (kernels +
    D2H of the records), and the q8 records against the f32 records field by field."""
    cfg = copy.deepcopy(config)
    cfg['signal_processing']['lstm_arith'] = 'f32'
    env_arith = os.environ.pop('PXG_LSTM_ARITH', None)
    try:
        c2 = N.NativeContext(cfg, device_id=local_rank)
    finally:
        if env_arith is not None:
            os.environ['PXG_LSTM_ARITH'] = env_arith
    try:
        c2.upload(base['arena'], base['offsets'], base['calib'], inject)
        buf = np.zeros(len(res_q8), dtype=N.RESULT_DTYPE)
        for _ in range(2):
            c2.run(mask)
            r32 = c2.download(buf)
        c2.sync()
        steps = max(3, min(args.steps, 5))
        acc = {k: 0.0 for k in N.TIMER_NAMES}
        t0 = time.perf_counter()
        for _ in range(steps):
            c2.run(mask)
            r32 = c2.download(buf)
            times, _ = c2.stage_times()
            for k in acc:
                acc[k] += times[k]
        c2.sync()
        wall = time.perf_counter() - t0
    finally:
        c2.close()
    both = (res_q8['status'] == 0) & (r32['status'] == 0)
    pushed = both & (res_q8['bc_pushed'] == 1) & (r32['bc_pushed'] == 1)
    return {
        'reads_per_s': len(res_q8) * steps / wall, 'ms_per_step': wall / steps * 1e3, 'steps': steps,
        'stage_ms': {k: round(v / steps, 4) for k, v in acc.items()},
        'kernel': 'k_scaler_lstm_q (v_mfma_f32_16x16x4_f32)',
        'roofline_frac_fp32_mfma': len(res_q8) * FLOP_SCALER / (acc['scaler_lstm'] / steps * 1e-3) / PEAK_FP32_MFMA
        if acc['scaler_lstm'] else None,
        'q8_speedup_scaler_lstm': (acc['scaler_lstm'] / steps) / stage_ms_q8['scaler_lstm'] if stage_ms_q8['scaler_lstm'] else None,
        # decisions of the two arithmetics on the SAME reads
        'q8_vs_f32': {
            'reads': int(len(res_q8)),
            'status_flips': int((res_q8['status'] != r32['status']).sum()),
            'reads_with_a_segment_boundary_moved': int(((res_q8['seg_first'] != r32['seg_first']) |
                                                        (res_q8['seg_last'] != r32['seg_last'])).any(1)[both].sum()),
            'window_gate_flips': int((res_q8['bc_pushed'] != r32['bc_pushed'])[both].sum()),
            'argmax_flips': int((res_q8['bc_label'] != r32['bc_label'])[pushed].sum()),
            'called_uncalled_flips': int((res_q8['bc_called'] != r32['bc_called'])[pushed].sum()),
            'phred_changes': int((res_q8['bc_phred'] != r32['bc_phred'])[pushed].sum()),
            'scale_max_abs_diff': float(np.abs(res_q8['scale'][both] - r32['scale'][both]).max()) if both.any() else None,
            'shift_max_abs_diff': float(np.abs(res_q8['shift'][both] - r32['shift'][both]).max()) if both.any() else None,
            'softmax_max_abs_diff': float(np.abs(res_q8['probs'][pushed] - r32['probs'][pushed]).max()) if pushed.any() else None,
            'polya_called_flips': int((res_q8['polya_called'] != r32['polya_called'])[both].sum()),
            'polya_interval_changes': int(((res_q8['polya_begin'] != r32['polya_begin']) |
                                           (res_q8['polya_end'] != r32['polya_end']))[both].sum()),
        },
    }


def latency_leg(ctx, base, inject, mask, n_small=1024, reps=4):
    """Small resident batches: the stage times of an `n_small`-read batch with the latency forms of
    K2 / K5a / K5b (k_lstm_q8_lat.hip: 4-read tiles on four times as many CUs, chosen by the launchers up to 8 x #CU
    reads) and with the 16-read-tile kernels forced on the same batch (PXG_K2_LAT_MAX = PXG_K5_LAT_MAX = 0); the
    records of the two runs must be identical."""
    n_small = min(n_small, len(base['offsets']) - 1)
    o = base['offsets']
    arena, off = base['arena'][:o[n_small]], o[:n_small + 1]
    cal = base['calib'][:n_small]
    ctx.upload(arena, off, cal, None if inject is None else inject[:n_small])
    out = {'reads': int(n_small), 'best_of': reps}
    recs = {}
    saved = {k: os.environ.get(k) for k in ('PXG_K2_LAT_MAX', 'PXG_K5_LAT_MAX')}
    try:
        for mode in ('tile_form', 'latency_form'):
            for k in saved:
                if mode == 'tile_form':
                    os.environ[k] = '0'
                else:
                    os.environ.pop(k, None)
            best = None
            for _ in range(reps):
                ctx.run(mask)
                ctx.sync()
                ms, _n = ctx.stage_times()
                if best is None or ms['total'] < best['total']:
                    best = ms
            recs[mode] = ctx.download()
            out[mode] = {k: round(best[k], 4) for k in ('scaler_lstm', 'segment', 'demux_bidir', 'demux_top', 'total')}
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    out['records_identical'] = bool(all(np.array_equal(recs['tile_form'][f], recs['latency_form'][f], equal_nan=True)
                                        for f in recs['tile_form'].dtype.names))
    out['speedup_K2'] = round(out['tile_form']['scaler_lstm'] / out['latency_form']['scaler_lstm'], 3) if out['latency_form']['scaler_lstm'] else None
    out['speedup_batch'] = round(out['tile_form']['total'] / out['latency_form']['total'], 3) if out['latency_form']['total'] else None
    out['K2_cycles_per_step_at_2.2GHz'] = round(out['latency_form']['scaler_lstm'] * 1e-3 / 2001 * 2.2e9)
    return out


def full_leg(args, ctx, base, lens, inject, orc, n_check=64, big_reads=100000):
    """BASELINE configs[3] inside the default line: the demux stages + poly(A) (K6) + the Guppy block means (K7a) + the
    pseudo-fusion window scan (K7b) on the SAME resident 10 000-read batch, timed like the headline (kernels + D2H of
    the records + the scan's candidate lists, every step), the records and the candidate lists of the first reads
    against the oracle, and once more at configs[3]'s stated batch size -- `big_reads` reads, the batch tiled on the
    device.  Per-kernel bounds: live HIP-event times beside the committed counters of the rocprofv3 passes of
    `--workload full` (static files, named)."""
    mask = N.STAGE_ALL_DEMUX | N.STAGE_POLYA
    n = len(lens)

    def measure(n_reads, lens_n, steps, warm=2):
        ev_first = np.zeros(n_reads, dtype=np.int64)
        ev_blocks = lens_n // 15
        buf = np.zeros(n_reads, dtype=N.RESULT_DTYPE)
        for _ in range(warm):
            ctx.run(mask)
            cand = ctx.unsplit_scan(ev_first, ev_blocks)
            res = ctx.download(buf)
        ctx.sync()
        acc = {k: 0.0 for k in N.TIMER_NAMES}
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.run(mask)
            cand = ctx.unsplit_scan(ev_first, ev_blocks)
            res = ctx.download(buf)
            times, _ = ctx.stage_times()
            for k in acc:
                acc[k] += times[k]
        ctx.sync()
        wall = time.perf_counter() - t0
        return wall, {k: v / steps for k, v in acc.items()}, res, cand, ev_blocks

    steps = max(3, min(args.steps, 10))
    wall, stage_ms, res, cand, ev_blocks = measure(n, lens, steps)
    out = {
        'config': 'BASELINE configs[3] stages (a1-a19) on the headline batch: {} reads x ~{} samples, resident'.format(n, args.samples),
        'reads_per_s': n * steps / wall, 'ms_per_step': wall / steps * 1e3, 'steps': steps,
        'stage_ms': {k: round(v, 4) for k, v in stage_ms.items()},
        # (since round 5 the scan's kernels run on a second stream beside K6: the stage times overlap, their sum is
        #  no longer the step's GPU time)
        'stage_ms_sum': round(stage_ms['total'] + stage_ms['event_means'] + stage_ms['unsplit'], 4),
        'polya_called': int((res['polya_called'] == 1).sum()),
        'reads_with_fusion_candidates': int((np.asarray(cand[1]) > 0).sum()),
        'statuses': {N.STATUS_NAMES[int(k)]: int(v) for k, v in zip(*np.unique(res['status'], return_counts=True))},
    }
    # ---- concordance, poly(A) and candidate lists included --------------------------------------------------
    if orc is not None:
        ns = min(n_check, n)
        parts = [base['arena'][base['offsets'][b]:base['offsets'][b + 1]] for b in range(ns)]
        s_arena, s_off = N.pack_reads(parts)
        s_cal = base['calib'][:ns]
        want = orc.process_batch(s_arena, s_off, s_cal, None if inject is None else inject[:ns], mask)
        got = res[:ns]
        same = [f for f in got.dtype.names if np.array_equal(got[f], want[f], equal_nan=True)]
        iv, cnt, start = cand
        cand_mismatch = 0
        for i in range(ns):
            w = want[i]
            if w['status'] != 0 or w['seg_first'][3] < 0 or ev_blocks[i] <= 0:
                cand_mismatch += int(cnt[i] != 0)
                continue
            _, sc = orc.guppy_event_means(parts[i], s_cal[i], 0, int(ev_blocks[i]), w['scale'], w['shift'])
            wiv, wc = orc.unsplit_scan(sc, 0, (int(w['seg_last'][3]) + 1) * 15, float(s_cal[i]['sampling_rate']))
            cand_mismatch += int(wc != cnt[i] or wiv.tolist() != iv[start[i]:start[i + 1]].tolist())
        out['concordance'] = {
            'reads_compared': ns, 'all_fields_bit_exact': len(same) == len(got.dtype.names),
            'fields_differing': [f for f in got.dtype.names if f not in same],
            'polya_called_mismatch': int((got['polya_called'] != want['polya_called']).sum()),
            'polya_interval_mismatch': int(((got['polya_begin'] != want['polya_begin']) |
                                            (got['polya_end'] != want['polya_end'])).sum()),
            'polya_tails_in_sample': int((want['polya_called'] == 1).sum()),
            'unsplit_candidate_mismatch': cand_mismatch,
            'checker': 'oracle/libpxo.so in the context\'s arithmetic: process_batch (all record fields) + '
                       'guppy_event_means / unsplit_scan (candidate lists)'}
    # ---- per-kernel bounds -----------------------------------------------------------------------------------
    n_scaled = int((res['status'] != N.STATUS_CODE['scaler_signal_too_short']).sum())
    roof = []
    dur = stage_ms['scaler_lstm'] * 1e-3
    if dur:
        roof.append({'kernel': 'k_scaler_lstm_q8 (K2)', 'bound': 'mfma', 'kernel_ms': round(stage_ms['scaler_lstm'], 4),
                     'achieved': n_scaled * OPS_SCALER_Q8 / dur / 1e12, 'unit': 'TOP/s (int8)',
                     'peak': PEAK_I8_MFMA / 1e12, 'frac': n_scaled * OPS_SCALER_Q8 / dur / PEAK_I8_MFMA,
                     'frac_of_measured_int8_ceiling': n_scaled * OPS_SCALER_Q8 / dur / PEAK_I8_MFMA_MEASURED})
    static = {}
    try:
        with open(os.path.join(ROOT, BOUNDS_FILE)) as fh:
            static = json.load(fh)['kernels']
    except (OSError, KeyError, ValueError):
        pass
    samples = float(lens.sum())
    blocks = float((lens // 15).sum())
    for name, timer, alg_bytes, what in (
            ('k_polya (K6)', 'polya', None, 'serial peak FSM + interval DP per read (16 lanes per read): issue / latency bound'),
            ('k_guppy_event_means (K7a)', 'event_means', samples * 2 + blocks * 4,
             'every int16 sample in once, one float32 block mean out; fp64 pA conversion: VALU issue at HBM rate'),
            ('k_unsplit_scan_w (K7b)', 'unsplit', None, 'fp64 Viterbi recurrence, one window per lane, ~6 overlapping windows per read: fp64 issue + LDS latency at one wave per SIMD')):
        ms = stage_ms[timer]
        row = {'kernel': name, 'bound': 'hbm' if alg_bytes else 'issue', 'kernel_ms': round(ms, 4), 'what': what}
        if alg_bytes and ms:
            row.update({'achieved': alg_bytes / (ms * 1e-3) / 1e9, 'unit': 'GB/s', 'peak': PEAK_HBM / 1e9,
                        'frac': alg_bytes / (ms * 1e-3) / PEAK_HBM, 'algorithmic_bytes': alg_bytes})
        key = name.split(' ')[0]
        if key in static:
            row['counters_static'] = {k: static[key][k] for k in (
                'avg_ms', 'valu_issue_frac_of_simd_cycles', 'wait_any_frac_of_wave_cycles',
                'lds_bank_conflict_frac_of_lds_cycles', 'hbm_bytes', 'hbm_GBps') if k in static[key]}
            row['counters_source'] = 'static: ' + BOUNDS_FILE
        roof.append(row)
    out['roofline'] = roof
    # ---- configs[3]'s stated batch size: tiled on the device -----------------------------------------------
    try:
        ctx.upload_tiled(big_reads, base['arena'], base['offsets'], base['calib'], inject, phase=0)
        lens_big = lens[np.arange(big_reads) % n]
        w2, st2, res2, cand2, _ = measure(big_reads, lens_big, 3, warm=1)
        out['at_{}_reads'.format(big_reads)] = {
            'reads_per_s': big_reads * 3 / w2, 'ms_per_step': w2 / 3 * 1e3, 'steps': 3,
            'distinct_reads': n, 'tiled_on_device': True,
            'stage_ms': {k: round(v, 4) for k, v in st2.items()},
            'records_equal_to_the_10000_read_run': bool(res2[:n].tobytes() == res.tobytes()),
            'polya_called': int((res2['polya_called'] == 1).sum())}
    except N.PxgError as exc:
        out['at_{}_reads'.format(big_reads)] = {'error': str(exc)}
    return out


def make_context(args, config, local_rank):
    if CONTEXT_CLASS is not None:
        return CONTEXT_CLASS(config, device_id=local_rank)
    return N.NativeContext(config, device_id=local_rank)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        respawn_ranks(args)
    # stdout carries exactly ONE line, the JSON: libraries print there too (RCCL writes a
    # version banner to stdout when the first communicator comes up), so fd 1 is pointed at
    # stderr for the whole run and the JSON goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = 0 if SHARE_GPU else int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus {} but WORLD_SIZE {} (launch with --nproc-per-node {} '
                         'or without a launcher)'.format(args.gpus, world, args.gpus))
    standin = CONTEXT_CLASS is not None
    config = default_config()
    if args.lstm_arith:
        os.environ.pop('PXG_LSTM_ARITH', None)
        config['signal_processing']['lstm_arith'] = args.lstm_arith
    arith = os.environ.get('PXG_LSTM_ARITH') or str(config['signal_processing'].get('lstm_arith', 'q8'))
    wl_cfg, wl_stages, wl_metric = STAGES[args.workload]
    mask = {'demux': N.STAGE_ALL_DEMUX, 'chimera': N.STAGE_ALL_DEMUX,
            'polya': N.STAGE_ALL_DEMUX | N.STAGE_POLYA, 'full': N.STAGE_ALL_DEMUX | N.STAGE_POLYA,
            'segment': N.STAGE_SEGMENT}[args.workload]
    use_inject = args.workload == 'segment'
    scan = args.workload in ('chimera', 'full')

    # ---- this rank's shard of ONE global run (reads are independent units: contiguous
    # blocks by rank; read_index is global) --------------------------------------
    if args.scaling == 'strong':
        total = args.total_reads
        lo, hi = shard_range(total, rank, world)
    else:
        total = args.reads * world
        lo, hi = rank * args.reads, (rank + 1) * args.reads
    n_local = hi - lo
    shard_sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0]
                   if args.scaling == 'strong' else args.reads for r in range(world)]
    n_base = args.base_reads if args.base_reads >= 0 else (0 if max(shard_sizes) <= 16384 else 2048)
    t_gen = time.perf_counter()
    if n_base:      # K distinct reads (same on every rank), global read i = base read i % K
        base = synth_batch(n_base, seed=args.seed, samples_per_read=args.samples, length_dist=args.length_dist)
        which = (lo + np.arange(n_local)) % n_base
    else:           # every read distinct: rank r draws its own block of the run
        base = synth_batch(n_local, seed=args.seed + 1000 * rank, samples_per_read=args.samples, length_dist=args.length_dist)
        which = np.arange(n_local)
    t_gen = time.perf_counter() - t_gen
    lens = np.diff(base['offsets'])[which]
    inject = base['scale_shift'] if use_inject else None

    # ---- CPU baseline over all cores: before anything touches the HIP runtime (fork) --------
    cpu_all = None
    if rank == 0 and world == 1 and args.cpu_all_cores_sample > 0 and not standin:
        cpu_all = cpu_all_cores(config, base, mask, use_inject, args.cpu_all_cores_sample)

    dist = None
    # PXG_BENCH_FORCE_DIST=1 takes the multi-rank path (torch + RCCL initialised before the
    # HIP library, collectives issued) with a single rank: the way the N>1 path is validated
    # on a 1-GPU box
    force_dist = os.environ.get('PXG_BENCH_FORCE_DIST') == '1'
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        if standin or SHARE_GPU:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))

    if args.end_to_end:
        return end_to_end(args, config, rank, local_rank, world, dist, standin, base, which,
                          lo, total, mask, json_fd)
    ctx = make_context(args, config, local_rank)
    info = ctx.device_info()
    # this rank's process (and, by first touch, the host memory it allocates from here on) goes
    # to the NUMA node of ITS GPU; the batch that exists already is copied there
    from poreplex_amd.distributed import bind_to_gpu_numa
    numa = {'numa_node': None, 'cpus': None, 'pci': None} if standin else bind_to_gpu_numa(local_rank)
    if numa['numa_node'] is not None:
        base['arena'] = base['arena'].copy()

    # --filter-chimera: Guppy block frame of every read (first sample 0, stride 15)
    ev_first = np.zeros(n_local, dtype=np.int64)
    ev_blocks = lens // 15

    def step():
        ctx.run(mask)
        if scan:
            ctx.unsplit_scan(ev_first, ev_blocks)
    t_up0 = time.perf_counter()
    if n_base:
        ctx.upload_tiled(n_local, base['arena'], base['offsets'], base['calib'], inject,
                         phase=lo % n_base)
    else:
        ctx.upload(base['arena'], base['offsets'], base['calib'], inject)
    t_upload = time.perf_counter() - t_up0

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
            if not standin:
                import torch
                torch.cuda.synchronize()

    # the result records land in ONE page-locked host buffer, reused by every step
    res_buf = N.page_exclusive(n_local, N.RESULT_DTYPE, fill=0)
    if not standin:
        ctx.pin(res_buf)
    for _ in range(args.warmup):
        step()
        res = ctx.download(res_buf)
        gather_labels(res, dist, first_index=lo, sizes=shard_sizes, force=force_dist)
    barrier()
    stage_acc = {k: 0.0 for k in N.TIMER_NAMES}
    t0 = time.perf_counter()
    pending, prev = None, None
    step_done = [t0]
    for _ in range(args.steps):
        step()                           # enqueue this step's kernels
        if prev is not None:
            # the label records of the PREVIOUS step are built, and their RCCL all-gather (N>1)
            # launched, on the host while this step's kernels run; the gather before that one is
            # collected first (asynchronous on RCCL's stream, one step of slack)
            if pending is not None:
                labels = pending()
            pending = gather_labels_start(prev, dist, first_index=lo, sizes=shard_sizes, force=force_dist)
        res = ctx.download(res_buf)      # D2H of the result records is part of a step
        step_done.append(time.perf_counter())
        prev = res
        times, _ = ctx.stage_times()
        for k in stage_acc:
            stage_acc[k] += times[k]
    if pending is not None:
        labels = pending()
    pending = gather_labels_start(prev, dist, first_index=lo, sizes=shard_sizes, force=force_dist)
    labels = pending()
    barrier()
    elapsed = time.perf_counter() - t0
    n_ranks = 1
    if dist is not None:
        import torch
        t = torch.tensor([elapsed, 1.0], dtype=torch.float64,
                         device=collective_device(standin))
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)     # ranks that took part, counted by RCCL
        elapsed, n_ranks = float(t[0].item()), int(round(t[1].item()))
    value = total * args.steps / elapsed

    # ---- legs every rank takes part in (N > 1: all ranks at the same time) -------------------
    pcie_local, pcie_error = {}, None
    if not n_base and not (args.no_overlap_test and world == 1):
        try:
            if args.no_overlap_test:
                raise N.PxgError('skipped (--no-overlap-test)')
            pcie_local = pcie_legs(ctx, base, inject, step, n_local, args, value / world)
        except N.PxgError as exc:
            pcie_error = str(exc)
    pcie_ranks = None
    if dist is not None:            # min / mean / max over the ranks, by collective
        keys = ('pcie_overlapped_reads_per_s', 'pcie_overlapped_encoded_reads_per_s', 'h2d_GBps')
        pcie_ranks = {k: rank_stats(pcie_local.get(k), dist, standin) for k in keys}
    configs4 = None
    if dist is not None and args.scaling == 'weak' and not args.no_configs4:
        configs4 = strong_leg(args, ctx, dist, rank, world, mask, standin, force_dist, barrier)
        # the weak batch again (the API / concordance legs below use it)
        if n_base:
            ctx.upload_tiled(n_local, base['arena'], base['offsets'], base['calib'], inject, phase=lo % n_base)
        else:
            ctx.upload(base['arena'], base['offsets'], base['calib'], inject)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    stage_ms = {k: v / args.steps for k, v in stage_acc.items()}
    n_ok = int((res['status'] == 0).sum())
    n_scaled = int((res['status'] != N.STATUS_CODE['scaler_signal_too_short']).sum())
    n_pushed = int(res['bc_pushed'].sum())
    truth_barcode = base['barcode'][which]

    # ---- roofline of the dominant kernel ------------------------------------
    if args.workload != 'segment':
        dur = stage_ms['scaler_lstm'] * 1e-3
        flops = n_scaled * FLOP_SCALER
        # k_lstm.hip pxg_launch_scaler_lstm: time-sliced kernel whenever the batch has more
        # 16-read tiles than the 2 x #CU resident workgroups
        if arith == 'q8':
            # executed on the int8 matrix pipe: `achieved` = algorithmic int8 digit products (8 per float32
            # multiply-add, no padding) / launch duration against the dense int8 MFMA peak; beside it the
            # executed rate (padding included) and what the same launch is worth in the network's own
            # float32 FLOPs against the fp32 MFMA roof the float32 arithmetic is bound by
            ops = n_scaled * OPS_SCALER_Q8
            roofline = {'kernel': 'k_scaler_lstm_q8', 'bound': 'mfma',
                        'achieved': ops / dur / 1e12 if dur else None, 'peak': PEAK_I8_MFMA / 1e12,
                        'unit': 'TOP/s (int8)', 'frac': ops / dur / PEAK_I8_MFMA if dur else None,
                        # the guide lists no int8 spec figure: `peak` is 2 x its dense bf16 figure (nominal, 2.4 GHz);
                        # beside it the fraction of the guide's MEASURED v_mfma_i32_16x16x64_i8 ceiling
                        'peak_guide_measured': PEAK_I8_MFMA_MEASURED / 1e12,
                        'frac_of_guide_measured_ceiling': ops / dur / PEAK_I8_MFMA_MEASURED if dur else None,
                        'traffic': None, 'algorithmic_int8_op_per_read': OPS_SCALER_Q8,
                        'executed_int8_TOPs': n_scaled * OPS_SCALER_Q8_EXECUTED / dur / 1e12 if dur else None,
                        'executed_frac': n_scaled * OPS_SCALER_Q8_EXECUTED / dur / PEAK_I8_MFMA if dur else None,
                        'algorithmic_fp32_TFLOPs': n_scaled * FLOP_SCALER / dur / 1e12 if dur else None,
                        'frac_of_fp32_mfma_peak': n_scaled * FLOP_SCALER / dur / PEAK_FP32_MFMA if dur else None,
                        'algorithmic_flop_per_read': FLOP_SCALER, 'kernel_ms': stage_ms['scaler_lstm'],
                        'note': 'per wave and step: 63 v_mfma_i32_16x16x64_i8 (72 until layer 2 lost its k padding, round 5; '
                                '~16.5 cycles each per SIMD) + ~440 VALU instructions at ~3.5 cycles: a SIMD issues one or the '
                                'other -- with two waves per SIMD the shadow of one wave\'s MFMAs is already filled by the '
                                'other wave\'s VALU (profiles/r05/ubench_i8_mfma_32x32.txt, ab_k2_sched.txt), the spline rows '
                                'cost 11 % and the step barrier 3 % (profiles/r04/k2_ablation.txt); DESIGN.md 3.1'}
        else:
            kernel = 'k_scaler_lstm_q' if (n_local + 15) // 16 > 2 * info['compute_units'] else 'k_scaler_lstm'
            roofline = {'kernel': kernel, 'bound': 'mfma',
                        'achieved': flops / dur / 1e12 if dur else None, 'peak': PEAK_FP32_MFMA / 1e12,
                        'unit': 'TFLOP/s', 'frac': flops / dur / PEAK_FP32_MFMA if dur else None,
                        'traffic': None, 'algorithmic_flop_per_read': FLOP_SCALER,
                        'kernel_ms': stage_ms['scaler_lstm']}
    else:
        dur = stage_ms['segment'] * 1e-3
        nbytes = float(np.minimum(lens, 100000).sum() * 2 + n_local * 88)
        roofline = {'kernel': 'k_viterbi_ltr', 'bound': 'hbm',
                    'achieved': nbytes / dur / 1e9 if dur else None,
                    'peak': PEAK_HBM / 1e9, 'unit': 'GB/s', 'frac': nbytes / dur / PEAK_HBM if dur else None,
                    'traffic': None, 'algorithmic_bytes_per_read': nbytes / n_local,
                    'kernel_ms': stage_ms['segment']}
    # the same fraction from the committed rocprofv3 --kernel-trace --stats pass of this command (average over
    # its launches, cold ones included): static, from profiles/; the live figure above is the HIP-event mean
    if n_local == 10000 and args.samples == 60000 and args.seed == 924 and not n_base and args.length_dist is None:
        try:
            import csv
            with open(os.path.join(ROOT, ROCPROF_STATS[args.workload if args.workload == 'full' else 'demux'])) as fh:
                row = next(r for r in csv.DictReader(fh) if roofline['kernel'] + '(' in r['Name'] or r['Name'].startswith(roofline['kernel']))
            avg_ms = float(row['AverageNs']) / 1e6
            roofline['rocprof_avg_ms'] = avg_ms
            roofline['rocprof_launches'] = int(row['Calls'])
            if roofline.get('frac') and roofline.get('kernel_ms'):
                roofline['frac_at_rocprof_avg'] = roofline['frac'] * roofline['kernel_ms'] / avg_ms
            roofline['rocprof_source'] = 'static: ' + ROCPROF_STATS[args.workload if args.workload == 'full' else 'demux']
            # the clock the kernel really ran at in the committed PMC pass (GRBM_GUI_ACTIVE / 8 XCDs / duration): the
            # nominal peaks are quoted at 2.4 GHz
            bounds_file = BOUNDS_FILE if args.workload == 'full' else BOUNDS_DEMUX_FILE
            with open(os.path.join(ROOT, bounds_file)) as fh:
                kb_ = json.load(fh)['kernels'].get(roofline['kernel'], {})
            if kb_.get('clock_GHz'):
                roofline['clock_GHz_in_profile'] = kb_['clock_GHz']
                roofline['clock_source'] = 'static: ' + bounds_file + ' (PMC pass of this workload: GRBM_GUI_ACTIVE / 8 XCDs / duration)'
                for key in ('valu_issue_frac_of_simd_cycles', 'mfma_busy_frac_of_simd_cycles', 'lds_busy_frac',
                            'lds_bank_conflict_frac_of_lds_cycles', 'wait_any_frac_of_wave_cycles'):
                    if key in kb_:
                        roofline.setdefault('counters_in_profile', {})[key] = kb_[key]
        except (OSError, KeyError, StopIteration, ValueError):
            pass
    # HBM bytes per launch of that kernel: NOT measured by this run (PMC counters need
    # rocprofv3); a static figure from the committed PMC passes of this exact default workload
    if n_local == 10000 and args.samples == 60000 and args.seed == 924 and not n_base:
        try:
            with open(os.path.join(ROOT, TRAFFIC_FILE)) as fh:
                roofline['traffic'] = json.load(fh)['kernels'][roofline['kernel']]['hbm_bytes']
            roofline['traffic_source'] = ('static: {} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '
                                          'this workload, tools/prof.sh), not re-measured here'
                                          .format(TRAFFIC_FILE))
        except (OSError, KeyError):
            pass
    # secondary figures for DESIGN.md (not part of the contract)
    alg_bytes = float(np.minimum(lens, 100000).sum() * 2 + n_local * 88)
    extra = {
        'vs_published_whole_pipeline': None if standin else value / PUBLISHED_READS_PER_S,     # (an anchor, not this metric: vs_baseline_basis)
        'stage_ms': {k: round(v, 4) for k, v in stage_ms.items()},
        # stage_ms['total'] spans pxg_batch_run only (K1 ... finalize, poly(A) included); the chimera filter's two
        # stages (event_means, unsplit) run in their own call and have their own timers -- behind the run, or, when the
        # run had the poly(A) stage, on a second stream BESIDE K6 (round 5): then the three figures overlap
        'gpu_ms_per_step': round(stage_ms['total'] + (0.0 if (scan and (mask & N.STAGE_POLYA)) else
                                                      stage_ms['event_means'] + stage_ms['unsplit']), 4),
        'scan_overlaps_polya': bool(scan and (mask & N.STAGE_POLYA)),
        # wall - GPU = the step's host share (D2H of the records, launches, the scan's one wait; with the overlap also
        # whatever of the scan outlasts the run)
        'host_ms_per_step': round(elapsed / args.steps * 1e3 - stage_ms['total']
                                  - (0.0 if (scan and (mask & N.STAGE_POLYA)) else
                                     stage_ms['event_means'] + stage_ms['unsplit']), 4),
        'hbm_frac_whole_path': value / world * (alg_bytes / n_local) / PEAK_HBM,
        'fp32_frac_whole_path': value / world * (FLOP_SCALER + FLOP_BIDIR + FLOP_TOP) / PEAK_FP32_MFMA,
        'upload_s': round(t_upload, 4), 'synth_s': round(t_gen, 2),
        'reads_ok': n_ok, 'reads_barcoded_window': n_pushed,
        # calls against the barcode the generator planted (synth.py; -1 = no barcode signal)
        'barcode_called': int((res['bc_called'] == 1).sum()),
        'barcode_called_correct': int(((res['bc_called'] == 1) & (res['bc_label'] == truth_barcode)).sum()),
        'barcode_planted': int((truth_barcode >= 0).sum()),
        # wall time of each step of the timed loop (records of step k on the host), this rank
        'step_ms': {k: round(float(f(np.diff(step_done))) * 1e3, 4)
                    for k, f in (('min', np.min), ('median', np.median), ('max', np.max))},
        'labels_gathered': int(len(labels)),
        'labels_read_index_unique': bool(len(np.unique(labels['read_index'])) == len(labels)),
        'ranks_counted_by_collective': n_ranks,
    }

    # ---- PCIe-inclusive rates (never `value`): (i) upload and compute alternating,
    # (ii) the double-buffered loader path: every step uploads a full batch from pinned host
    # memory on the copy stream while the previous one computes (pxg_batch_stage / swap),
    # (iii) the same with encoded samples decoded on the device -- measured above on every rank
    if not n_base and not standin:
        extra['pcie_inclusive_reads_per_s'] = n_local / (elapsed / args.steps + t_upload)
    extra.update(pcie_local)
    if pcie_error is not None:
        extra['pcie_overlapped_reads_per_s'] = None
        extra['pcie_overlapped_error'] = pcie_error
    if pcie_ranks is not None:
        extra['pcie_over_ranks'] = pcie_ranks      # N > 1: all ranks staged at the same time
    if configs4 is not None:
        extra['configs4_strong'] = configs4
    extra['numa'] = numa

    # ---- CPU baseline + concordance: the oracle, rank 0, bounded sample -------
    cpu = None
    concordance = None
    if args.cpu_sample > 0 and not standin:
        from oracle.pxo import Oracle
        # cpu_baseline is the oracle in FLOAT32 arithmetic -- one fma chain per gate is what a CPU's SIMD
        # units run fastest (the q8 digit arithmetic is shaped for the int8 matrix pipe and costs the oracle
        # about twice the time: timing THAT would flatter the GPU); the concordance check below uses the
        # oracle in the arithmetic the GPU context runs
        cfg_f32 = copy.deepcopy(config)
        cfg_f32['signal_processing']['lstm_arith'] = 'f32'
        env_arith = os.environ.pop('PXG_LSTM_ARITH', None)
        orc_f32 = Oracle(cfg_f32)
        config_c = copy.deepcopy(config)
        config_c['signal_processing']['lstm_arith'] = arith
        orc = orc_f32 if arith == 'f32' else Oracle(config_c)
        if env_arith is not None:
            os.environ['PXG_LSTM_ARITH'] = env_arith
        # N > 1: no cpu_baseline (it is an N = 1 figure), only the concordance check on a small sample
        ns = min(args.cpu_sample if world == 1 else min(args.cpu_sample, 128), n_local)
        parts = [base['arena'][base['offsets'][b]:base['offsets'][b + 1]] for b in which[:ns]]
        s_arena, s_off = N.pack_reads(parts)
        s_cal = base['calib'][which[:ns]]
        inj = None if inject is None else inject[which[:ns]]
        c0 = time.perf_counter()
        want = orc_f32.process_batch(s_arena, s_off, s_cal, inj, mask)
        cpu_core_s = time.perf_counter() - c0
        if orc is not orc_f32:                # the checker, in the GPU's arithmetic (half the sample: it is slower)
            ns_c = max(1, ns // 2)
            want = orc.process_batch(s_arena[:s_off[ns_c]], s_off[:ns_c + 1], s_cal[:ns_c],
                                     None if inj is None else inj[:ns_c], mask)
        else:
            ns_c = ns
        cand_mismatch = None
        if scan:
            iv, cnt, start = ctx.unsplit_scan(ev_first, ev_blocks)
            cand_mismatch = 0
            for i in range(ns_c):
                w = want[i]
                if w['status'] != 0 or w['seg_first'][3] < 0 or ev_blocks[i] <= 0:
                    cand_mismatch += int(cnt[i] != 0)
                    continue
                _, sc = orc.guppy_event_means(parts[i], s_cal[i], 0, int(ev_blocks[i]), w['scale'], w['shift'])
                wiv, wc = orc.unsplit_scan(sc, 0, (int(w['seg_last'][3]) + 1) * 15,
                                           float(s_cal[i]['sampling_rate']))
                cand_mismatch += int(wc != cnt[i] or wiv.tolist() != iv[start[i]:start[i + 1]].tolist())
        cpu_s = cpu_core_s
        model, physical, usable, quota = host_description()
        cpu = {'value': ns / cpu_s, 'unit': 'reads/s', 'cores': 1, 'kind': 'port',
               'sample': 'first {} reads of the same batch, same stages, oracle/libpxo.so '
                         '(C restatement, gcc -O2 AVX2, float32 LSTM arithmetic), single thread, {:.1f} s'.format(ns, cpu_s),
               'arith': 'f32',
               'host_cpu': model, 'physical_cores': physical, 'usable_cores': usable,
               'cgroup_cpu_quota': quota,
               # what perfect scaling of the one-core rate over every physical core would give
               'ideal_all_cores': ns / cpu_s * physical,
               'speedup_vs_ideal_all_cores': value / world / (ns / cpu_s * physical),
               'speedup_vs_one_core': value / world / (ns / cpu_s)}
        if cpu_all is not None:
            cpu['all_cores'] = dict(cpu_all, sample='{} reads over {} worker processes (one per physical '
                                    'core, read shards; ProcessPoolExecutor shape of pipeline.py:96), '
                                    '{} s wall'.format(cpu_all['reads'], cpu_all['cores'], cpu_all['wall_s']))
            cpu['speedup_vs_all_cores'] = value / world / cpu_all['value']
        got = res[:ns_c]
        same = [f for f in got.dtype.names if np.array_equal(got[f], want[f], equal_nan=True)]
        concordance = {
            'reads_compared': ns_c, 'arith': arith,
            'status_mismatch': int((got['status'] != want['status']).sum()),
            'segment_mismatch': int(((got['seg_first'] != want['seg_first']) |
                                     (got['seg_last'] != want['seg_last'])).any(1).sum()),
            'barcode_label_mismatch': int((got['bc_label'] != want['bc_label']).sum()),
            'barcode_call_mismatch': int((got['bc_called'] != want['bc_called']).sum()),
            'softmax_max_abs_diff': float(np.abs(got['probs'] - want['probs']).max()),
            'all_fields_bit_exact': len(same) == len(got.dtype.names),
        }
        if cand_mismatch is not None:
            concordance['unsplit_candidate_mismatch'] = cand_mismatch
        concordance.update(unpinned_rows_block())
        if world > 1:
            cpu = None

    # ---- a run-shaped workload: the lengths a flow cell produces (9 000 ... 1 000 000 samples) instead of the
    # uniform ~60 000 of the headline; 1 024 distinct reads tiled on the device to the same 10 000-read batch ----
    if not standin and world == 1 and not n_base and not args.no_run_shaped_leg and args.length_dist is None and \
            args.workload in ('demux', 'full', 'polya') and not use_inject:
        try:
            extra['run_shaped'] = run_shaped_leg(args, ctx, mask, n_local, config=config, local_rank=local_rank)
            ctx.upload(base['arena'], base['offsets'], base['calib'], inject)        # the headline batch again
        except Exception as exc:                       # reported, never hidden
            extra['run_shaped'] = {'error': '{}: {}'.format(type(exc).__name__, exc)}

    # ---- the same batch in the OTHER arithmetic (float32 fma chains on the fp32 MFMA, rounds 1-3): its
    # rate, and every decision the two arithmetics take differently over the whole batch ----------------
    if arith == 'q8' and not standin and world == 1 and not n_base and not args.no_f32_leg and args.workload != 'segment':
        try:
            extra['f32_arith'] = f32_leg(args, config, local_rank, base, inject, mask, res, stage_ms)
        except Exception as exc:                       # reported, never hidden
            extra['f32_arith'] = {'error': '{}: {}'.format(type(exc).__name__, exc)}

    # ---- BASELINE configs[3] (a1-a19: + poly(A) + Guppy block means + window scan) in THIS line: the stages the
    # `full` workload times, on the headline batch and at 100 000 reads tiled on the device ---------------------
    if not standin and world == 1 and not n_base and not args.no_full_leg and args.workload == 'demux' and \
            args.length_dist is None and not use_inject:
        try:
            extra['full'] = full_leg(args, ctx, base, lens, inject, orc if (args.cpu_sample > 0) else None)
        except Exception as exc:                       # reported, never hidden
            extra['full'] = {'error': '{}: {}'.format(type(exc).__name__, exc)}
        ctx.upload(base['arena'], base['offsets'], base['calib'], inject)        # the headline batch again

    # ---- the reference-shaped API (never `value` of the default line): process_batch calls ----
    api = None
    if not standin and world == 1 and not n_base and not use_inject and \
            (args.api == 'process_batch' or not args.no_api_leg):
        try:
            api = {'raw': process_batch_leg(args, base, which, lo, mask, local_rank, False, res)}
            api['encoded'] = process_batch_leg(args, base, which, lo, mask, local_rank, True, res)
            extra['process_batch_reads_per_s'] = api['raw']['reads_per_s']
            extra['process_batch_encoded_bundle_reads_per_s'] = api['encoded']['reads_per_s']
            extra['process_batch'] = api
            if args.api != 'process_batch' and len(which) >= 128:
                # the reference's OWN batch size (commandline.py:402: 128 reads per call) from 32 worker threads:
                # calls that meet in the pipeline run as one GPU batch (include/pxg.h "small calls share a batch")
                small = copy.copy(args)
                small.in_flight, small.api_calls = 32, 320
                api['reference_batch_size_128'] = process_batch_leg(small, base, which[:128], lo, mask, local_rank, False, res[:128])
                extra['process_batch_128_read_calls_reads_per_s'] = api['reference_batch_size_128']['reads_per_s']
                try:            # ... and over the reference's input: the reads in a multi-read FAST5 file, no bundle
                    api['reference_batch_size_128']['from_fast5'] = fast5_calls_leg(args, base, which, lo, mask, local_rank, res)
                except Exception as exc:                   # reported, never hidden
                    api['reference_batch_size_128']['from_fast5'] = {'error': '{}: {}'.format(type(exc).__name__, exc)}
                if not args.no_worker_processes_leg:
                    try:
                        api['reference_batch_size_128']['worker_processes'] = worker_processes_leg(args, base, which, lo, mask, local_rank)
                    except Exception as exc:               # reported, never hidden
                        api['reference_batch_size_128']['worker_processes'] = {'error': '{}: {}'.format(type(exc).__name__, exc)}
        except Exception as exc:                       # reported, never hidden
            extra['process_batch_reads_per_s'] = None
            extra['process_batch_error'] = '{}: {}'.format(type(exc).__name__, exc)
            api = None

    if not standin and world == 1 and not args.no_fast5_leg:
        try:
            extra['fast5_ingest'] = fast5_ingest_leg(args, base, which)
        except Exception as exc:                       # reported, never hidden
            extra['fast5_ingest'] = {'error': '{}: {}'.format(type(exc).__name__, exc)}

    if args.workload in ('full', 'polya', 'chimera'):
        # what bounds the kernels beside K2 in this workload: counters of the committed PMC passes of
        # `--workload full` (static, tools/prof.sh; derived figures in the file named)
        try:
            with open(os.path.join(ROOT, BOUNDS_FILE)) as fh:
                kb = json.load(fh)['kernels']
            extra['kernel_bounds'] = {
                'source': 'static: {} (rocprofv3 PMC passes of --workload full, scan and poly(A) behind the run: PXG_NO_SCAN_OVERLAP=1 PXG_NO_POLYA_OVERLAP=1)'.format(BOUNDS_FILE),
                'k_polya': dict(kb['k_polya'], bound='latency / divergence of a per-read FSM: 40 % of the SIMD cycles issue '
                                'VALU, a third of the instructions are scalar, waves wait 45 % of their cycles'),
                'k_unsplit_scan_w': dict(kb.get('k_unsplit_scan_w') or kb['k_unsplit_scan'], bound='one window per lane (round 5): fp64 issue + LDS round trips at one '
                                       'wave per SIMD (the 8-lane kernel of round 4: 70 % VALU issue, 2.95 ms)'),
                'k_guppy_event_means': dict(kb['k_guppy_event_means'], bound='VALU issue 98 % (fp64 pA conversion, median-of-5 '
                                            'network) at 3.1 TB/s of HBM traffic'),
            }
        except (OSError, KeyError, ValueError):
            pass

    if not standin and world == 1 and not args.no_e2e_leg and not n_base and not use_inject and \
            args.workload == 'demux' and not SHARE_GPU and not force_dist:
        extra['end_to_end'] = end_to_end_legs(args)

    # ---- what the driver keeps: `parsed.roofline` whole (it drops `extra`).  Everything somebody needs to answer "what
    # did configs[3] do, which kernel is furthest from its roof, what does a small batch get" goes in here; LIVE
    # figures (this box, HIP events of this run) and STATIC ones (committed rocprofv3 files) in separate sub-dicts ----
    if roofline is not None and not standin and world == 1 and args.workload == 'demux':
        live = {'source': 'HIP events on the context stream, this run, this box', 'kernel_ms': roofline.get('kernel_ms'),
                'achieved': roofline.get('achieved'), 'frac': roofline.get('frac'), 'device': info['name'],
                'clock_khz_reported': info.get('clock_khz')}
        static = {k: roofline.get(k) for k in ('rocprof_avg_ms', 'rocprof_launches', 'frac_at_rocprof_avg', 'rocprof_source',
                                               'clock_GHz_in_profile', 'clock_source', 'counters_in_profile', 'traffic', 'traffic_source') if k in roofline}
        roofline['live'], roofline['static'] = live, static
        n_demux = int(res['bc_pushed'].sum())
        kern = []

        def krow(name, ms, bound, work, peak, unit, what):
            row = {'name': name, 'ms': round(ms, 4), 'bound': bound, 'what': what, 'counters_source': 'live HIP events (this run)'}
            if ms and work:
                row.update({'achieved': work / (ms * 1e-3) / (1e9 if bound == 'hbm' else 1e12), 'peak': peak / (1e9 if bound == 'hbm' else 1e12),
                            'unit': unit, 'frac': work / (ms * 1e-3) / peak})
            kern.append(row)
        head_bytes = float(np.minimum(lens, 30000).sum() * 2 + n_local * 2000 * 4)
        krow('k_head_pool (K1)', stage_ms['head_pool'], 'hbm', head_bytes, PEAK_HBM, 'GB/s', 'first 30 000 samples in, 2 000 block means out')
        krow('k_scaler_lstm_q8 (K2)', stage_ms['scaler_lstm'], 'mfma', n_scaled * OPS_SCALER_Q8 if arith == 'q8' else n_scaled * FLOP_SCALER,
             PEAK_I8_MFMA if arith == 'q8' else PEAK_FP32_MFMA, 'TOP/s (int8)' if arith == 'q8' else 'TFLOP/s', 'scaler LSTM: the dominant kernel')
        krow('k_viterbi_ltr (K3)', stage_ms['segment'], 'hbm', float(np.minimum(lens, 100000).sum() * 2 + n_local * 88), PEAK_HBM, 'GB/s',
             'pool + scale + 6-state Viterbi + run-length summary: fp64 VALU issue (producers + recurrence), not HBM')
        krow('k_barcode_window_raw (K4)', stage_ms['barcode_window'], 'hbm', float(n_local * (300 * 15 * 2 + 1200)), PEAK_HBM, 'GB/s', 'wave per read: latency')
        krow('k_demux_bidir_q8 (K5a)', stage_ms['demux_bidir'], 'mfma', n_demux * Q8_PRODUCTS * 2.0 * 300 * 2 * (48 * 192) if arith == 'q8' else n_demux * FLOP_BIDIR,
             PEAK_I8_MFMA if arith == 'q8' else PEAK_FP32_MFMA, 'TOP/s (int8)' if arith == 'q8' else 'TFLOP/s', 'bidirectional layer of the demux net')
        krow('k_demux_top_q8 (K5b)', stage_ms['demux_top'], 'mfma', n_demux * Q8_PRODUCTS * 2.0 * 300 * (160 * 256) if arith == 'q8' else n_demux * FLOP_TOP,
             PEAK_I8_MFMA if arith == 'q8' else PEAK_FP32_MFMA, 'TOP/s (int8)' if arith == 'q8' else 'TFLOP/s', 'top cell + dense + softmax')
        try:            # the committed counters of each kernel beside its live time (static, named)
            with open(os.path.join(ROOT, BOUNDS_DEMUX_FILE)) as fh:
                kb_all = json.load(fh)['kernels']
            for row in kern:
                st = kb_all.get(row['name'].split(' ')[0])
                if st:
                    row['static'] = {k: st[k] for k in ('avg_ms', 'clock_GHz', 'valu_issue_frac_of_simd_cycles', 'mfma_busy_frac_of_simd_cycles',
                                                        'lds_bank_conflict_frac_of_lds_cycles', 'wait_any_frac_of_wave_cycles', 'hbm_GBps') if k in st}
                    row['static_source'] = BOUNDS_DEMUX_FILE
        except (OSError, KeyError, ValueError):
            pass
        roofline['kernels'] = kern
        fracs = [(k['frac'], k['name']) for k in kern if k.get('frac') and k['ms'] >= 0.03 * stage_ms['total']]     # (kernels worth >= 3 % of the step)
        if fracs:
            roofline['furthest_from_its_roof'] = min(fracs)[1]
        if isinstance(extra.get('full'), dict) and 'reads_per_s' in extra['full']:
            fl = extra['full']
            roofline['full'] = {'config': 'BASELINE configs[3]: + poly(A) (K6) + Guppy block means (K7a) + pseudo-fusion scan (K7b), same resident batch',
                                'reads_per_s': fl['reads_per_s'], 'ms_per_step': fl['ms_per_step'],
                                'stage_ms': {k: fl['stage_ms'].get(k) for k in ('polya', 'event_means', 'unsplit', 'finalize', 'total')},
                                'at_100000_reads': fl.get('at_100000_reads'),
                                'concordance_ok': (fl.get('concordance') or {}).get('all_fields_bit_exact') if fl.get('concordance') else None,
                                'unsplit_candidate_mismatch': (fl.get('concordance') or {}).get('unsplit_candidate_mismatch')}
        try:
            if args.no_latency_leg:
                raise N.PxgError('skipped (--no-latency-leg)')
            lat = latency_leg(ctx, base, inject, mask)
            if api is not None and 'reference_batch_size_128' in api:
                r128 = api['reference_batch_size_128']
                lat['process_batch_128_read_calls'] = {'reads_per_s_32_threads': r128['reads_per_s'],
                                                       'reads_per_s_one_call_at_a_time': r128['one_call_at_a_time_reads_per_s'],
                                                       'mean_phase_ms_per_call': r128.get('mean_phase_ms_per_call'),
                                                       'merge_stats': r128.get('merge_stats'),
                                                       # (of the timed calls: judged + reported in one C pass)
                                                       'calls_on_the_plain_run_path': r128.get('calls_on_the_plain_run_path'),
                                                       'barcode_or_status_mismatch_vs_resident_records':
                                                           r128.get('barcode_or_status_mismatch_vs_resident_records'),
                                                       'worker_processes': r128.get('worker_processes'),
                                                       # the same calls with the reads in a multi-read FAST5 file (no bundle)
                                                       'from_fast5': r128.get('from_fast5')}
            roofline['latency_form'] = lat
        except Exception as exc:                       # reported, never hidden
            roofline['latency_form'] = {'error': '{}: {}'.format(type(exc).__name__, exc)}

    line = {
        'metric': wl_metric,
        'value': None if standin else value, 'unit': 'reads/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True, 'scaling': args.scaling,
        # no published number exists for THIS metric (BASELINE.json `published` is empty; BASELINE.md's ~230 reads/s is
        # Poreplex 0.1's whole pipeline incl. FAST5 I/O on 20 Xeon cores): null, the ratio to that anchor is in `extra`
        'vs_baseline': None,
        'vs_baseline_basis': 'no published figure for the hot path; extra.vs_published_whole_pipeline = value / 230 reads/s '
                             '(BASELINE.md section 1: Poreplex 0.1 whole pipeline, 20 Xeon cores), an order-of-magnitude '
                             'anchor only; see cpu_baseline for the same stages on this host',
        'dtype': DTYPE[arith],
        'data': 'TEST-STANDIN (no GPU work timed)' if standin else ('synthetic (ranks SHARING one GPU: a plumbing check, not a scaling number)' if SHARE_GPU else 'synthetic'),
        'config': {'workload': 'BASELINE configs[{}]: {} x ~{} int16 samples, stages {}'.format(
                       4 if args.scaling == 'strong' else wl_cfg,
                       '{} reads sharded over {} GPU(s)'.format(total, world) if args.scaling == 'strong'
                       else '{} reads/GPU'.format(args.reads), args.samples, wl_stages),
                   'reads_per_gpu': shard_sizes, 'total_reads_per_step': total,
                   'samples_per_read': args.samples,
                   'distinct_reads': n_base or n_local, 'tiled_on_device': bool(n_base),
                   'parallelism': 'reads sharded x{} (contiguous read_index blocks)'.format(world),
                   'device': info['name'], 'arch': info['arch'], 'compute_units': info['compute_units']},
        'roofline': roofline, 'cpu_baseline': cpu, 'concordance': concordance, 'extra': extra,
    }
    if args.api == 'process_batch' and api is not None:
        best = api['encoded'] if api['encoded']['reads_per_s'] > api['raw']['reads_per_s'] else api['raw']
        line.update({
            'metric': wl_metric + ' through process_batch(batchid, reads, config) -> result dicts',
            'value': api['raw']['reads_per_s'], 'steps': api['raw']['calls'], 'warmup': 3,
            'ms_per_step': api['raw']['ms_per_call'],
            'vs_baseline': None,
            'roofline': None})
        line['extra']['vs_published_whole_pipeline'] = api['raw']['reads_per_s'] / PUBLISHED_READS_PER_S
        line['config']['workload'] += ('; API leg: {} calls of {} reads from an int16 read bundle, {} in flight '
                                       '(best of raw / encoded bundle: {:.0f} reads/s)'.format(
                                           api['raw']['calls'], api['raw']['reads_per_call'],
                                           api['raw']['in_flight'], best['reads_per_s']))
    os.write(json_fd, (json.dumps(line) + '\n').encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
