#!/usr/bin/env python
"""bench.py -- reads/s of the raw-signal hot path (segment + barcode) on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W; for N>1 it
is launched by torch.distributed.run, one rank per GPU.  A "step" is one pass
of the hot path (head pool -> scaler LSTM -> pool+scale+Viterbi -> barcode
window -> demux LSTMs -> result records) over one resident batch of synthetic
reads; inputs are in HBM before the timed region.  Rank 0 prints ONE JSON line.

Workload = BASELINE.json configs[2] (the config the metric "reads/s
(segment+barcode)" is quoted on): a 10 000-read batch of ~60 000-sample reads
per GPU, all stages a1-a13.  `--workload segment` runs configs[1] instead.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from poreplex_amd import native as N  # noqa: E402
from poreplex_amd.config import default_config  # noqa: E402
from poreplex_amd.synth import synth_batch  # noqa: E402

# algorithmic work per read (SURVEY.md 8d / DESIGN.md "Roofline accounting")
FLOP_SCALER = 2.0 * 2000 * (48 * 192 + 96 * 192) + 2000 * 192 * 2 + 2 * 48 * 2   # ~111.4 MFLOP
FLOP_BIDIR = 2.0 * 300 * 2 * (48 * 192) + 300 * 2 * 192 * 2
FLOP_TOP = 2.0 * 300 * (160 * 256) + 2 * 64 * 5
PEAK_FP32_MFMA = 157.3e12      # MI355X_MICROARCH.md: dense fp32 MFMA peak, FLOP/s
PEAK_HBM = 8.0e12              # B/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--reads', type=int, default=10000, help='reads per GPU per step')
    ap.add_argument('--samples', type=int, default=60000, help='nominal samples per read')
    ap.add_argument('--workload', choices=['demux', 'segment', 'polya', 'chimera', 'full'], default='demux')
    ap.add_argument('--cpu-sample', type=int, default=1024,
                    help='reads timed on the host for cpu_baseline (0 = skip)')
    ap.add_argument('--check', type=int, default=64, help='reads compared with the oracle')
    ap.add_argument('--seed', type=int, default=924)
    ap.add_argument('--no-overlap-test', action='store_true',
                    help='skip the extra PCIe-overlapped steps (profiling runs: keeps the kernel '
                         'statistics to the timed steps)')
    return ap.parse_args()


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON: libraries print there too (RCCL writes a
    # version banner to stdout when the first communicator comes up), so fd 1 is pointed at
    # stderr for the whole run and the JSON goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    # PXG_BENCH_FORCE_DIST=1 takes the multi-rank path (torch + RCCL initialised before the
    # HIP library, collectives issued) with a single rank: the way the N>1 path is validated
    # on a 1-GPU box
    force_dist = os.environ.get('PXG_BENCH_FORCE_DIST') == '1'
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus {} but WORLD_SIZE {}'.format(args.gpus, world), file=sys.stderr)

    config = default_config()
    ctx = N.NativeContext(config, device_id=local_rank)
    info = ctx.device_info()

    # every rank draws its own shard of the synthetic run (reads are independent
    # units: contiguous blocks by rank)
    batch = synth_batch(args.reads, seed=args.seed + 1000 * rank, samples_per_read=args.samples)
    if args.workload == 'demux':
        mask, inject = N.STAGE_ALL_DEMUX, None
    elif args.workload == 'polya':
        mask, inject = N.STAGE_ALL_DEMUX | N.STAGE_POLYA, None
    elif args.workload == 'chimera':
        mask, inject = N.STAGE_ALL_DEMUX, None
    elif args.workload == 'full':            # configs[3]: + poly(A) + pseudo-fusion scan
        mask, inject = N.STAGE_ALL_DEMUX | N.STAGE_POLYA, None
    else:
        mask, inject = N.STAGE_SEGMENT, batch['scale_shift']
    # --filter-chimera: Guppy block frame of every read (first sample 0, stride 15)
    ev_first = np.zeros(args.reads, dtype=np.int64)
    ev_blocks = np.diff(batch['offsets']) // 15

    def step():
        ctx.run(mask)
        if args.workload in ('chimera', 'full'):
            ctx.unsplit_scan(ev_first, ev_blocks)
    t_up0 = time.perf_counter()
    ctx.upload(batch['arena'], batch['offsets'], batch['calib'], inject)
    t_upload = time.perf_counter() - t_up0

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    from poreplex_amd.distributed import gather_labels, gather_labels_start
    shard_sizes = [args.reads] * world          # static sharding: every rank owns args.reads reads
    for _ in range(args.warmup):
        step()
        res = ctx.download()
        gather_labels(res, dist, sizes=shard_sizes, force=force_dist)
    barrier()
    stage_acc = {k: 0.0 for k in N.TIMER_NAMES}
    t0 = time.perf_counter()
    pending = None
    for _ in range(args.steps):
        step()
        if pending is not None:          # last step's labels arrive while this step runs
            labels = pending()
        res = ctx.download()             # D2H of the result records is part of a step
        # RCCL all-gather of the label records (N>1), asynchronous on RCCL's stream
        pending = gather_labels_start(res, dist, sizes=shard_sizes, force=force_dist)
        times, _ = ctx.stage_times()
        for k in stage_acc:
            stage_acc[k] += times[k]
    labels = pending()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_reads = args.reads * world * args.steps
    value = total_reads / elapsed

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    stage_ms = {k: v / args.steps for k, v in stage_acc.items()}
    n_ok = int((res['status'] == 0).sum())
    n_scaled = int((res['status'] != N.STATUS_CODE['scaler_signal_too_short']).sum())
    n_pushed = int(res['bc_pushed'].sum())

    # ---- roofline of the dominant kernel ------------------------------------
    if args.workload in ('demux', 'polya', 'chimera', 'full'):
        dur = stage_ms['scaler_lstm'] * 1e-3
        flops = n_scaled * FLOP_SCALER
        tiles, slots = (n_scaled + 15) // 16, 2 * info['compute_units']
        roofline = {'kernel': 'k_scaler_lstm_q' if slots < (args.reads + 15) // 16 < 2 * slots
                    else 'k_scaler_lstm', 'bound': 'mfma',
                    'achieved': flops / dur / 1e12, 'peak': PEAK_FP32_MFMA / 1e12,
                    'unit': 'TFLOP/s', 'frac': flops / dur / PEAK_FP32_MFMA, 'traffic': None,
                    'algorithmic_flop_per_read': FLOP_SCALER}
    else:
        dur = stage_ms['segment'] * 1e-3
        nbytes = float(np.minimum(np.diff(batch['offsets']), 100000).sum() * 2 + args.reads * 88)
        roofline = {'kernel': 'k_viterbi_ltr', 'bound': 'hbm', 'achieved': nbytes / dur / 1e9,
                    'peak': PEAK_HBM / 1e9, 'unit': 'GB/s', 'frac': nbytes / dur / PEAK_HBM,
                    'traffic': None, 'algorithmic_bytes_per_read': nbytes / args.reads}
    # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes
    # (profiles/r01/i_final_hbm_traffic.json, collected with tools/prof.sh on this exact
    # default workload); None for any other workload size
    if args.reads == 10000 and args.samples == 60000 and args.seed == 924:
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01', 'i_final_hbm_traffic.json')) as fh:
                roofline['traffic'] = json.load(fh)['kernels'][roofline['kernel']]['hbm_bytes']
            roofline['traffic_source'] = 'profiles/r01/i_final_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE)'
        except (OSError, KeyError):
            pass
    # secondary figures for DESIGN.md (not part of the contract)
    alg_bytes = float(np.minimum(np.diff(batch['offsets']), 100000).sum() * 2 + args.reads * 88)
    extra = {
        'stage_ms': {k: round(v, 4) for k, v in stage_ms.items()},
        'hbm_frac_whole_path': value / world * (alg_bytes / args.reads) / PEAK_HBM,
        'fp32_frac_whole_path': value / world * (FLOP_SCALER + FLOP_BIDIR + FLOP_TOP) / PEAK_FP32_MFMA,
        'upload_s': round(t_upload, 4),
        'pcie_inclusive_reads_per_s': args.reads / (elapsed / args.steps + t_upload),
        'reads_ok': n_ok, 'reads_barcoded_window': n_pushed,
        # calls against the barcode the generator planted (synth.py; -1 = no barcode signal)
        'barcode_called': int((res['bc_called'] == 1).sum()),
        'barcode_called_correct': int(((res['bc_called'] == 1) &
                                       (res['bc_label'] == batch['barcode'])).sum()),
        'barcode_planted': int((batch['barcode'] >= 0).sum()),
        'labels_gathered': int(len(labels)),
    }

    # ---- PCIe-inclusive rate with the double-buffered loader path (not `value`): every step
    # uploads a full batch from pinned host memory on the copy stream while the previous one
    # computes (pxg_batch_stage / pxg_batch_swap)
    try:
        if args.no_overlap_test:
            raise N.PxgError('skipped (--no-overlap-test)')
        ctx.pin(batch['arena'])
        n_over = min(args.steps, 5)
        ctx.sync()
        o0 = time.perf_counter()
        for _ in range(n_over):
            step()
            ctx.stage(batch['arena'], batch['offsets'], batch['calib'], inject)
            ctx.download()
            ctx.swap()
        ctx.sync()
        extra['pcie_overlapped_reads_per_s'] = args.reads * n_over / (time.perf_counter() - o0)
        ctx.unpin(batch['arena'])
    except N.PxgError as exc:
        extra['pcie_overlapped_reads_per_s'] = None
        extra['pcie_overlapped_error'] = str(exc)

    # ---- CPU baseline + concordance: the oracle, rank 0, bounded sample -------
    cpu = None
    concordance = None
    if args.cpu_sample > 0:
        from oracle.pxo import Oracle
        orc = Oracle(config)
        ns = min(args.cpu_sample, args.reads)
        o = batch['offsets'][:ns + 1]
        inj = None if inject is None else inject[:ns]
        c0 = time.perf_counter()
        want = orc.process_batch(batch['arena'][:o[-1]], o, batch['calib'][:ns], inj, mask)
        cand_mismatch = None
        if args.workload in ('chimera', 'full'):
            iv, cnt = ctx.unsplit_scan(ev_first, ev_blocks)
            cand_mismatch = 0
            for i in range(ns):
                w = want[i]
                if w['status'] != 0 or w['seg_first'][3] < 0 or ev_blocks[i] <= 0:
                    cand_mismatch += int(cnt[i] != 0)
                    continue
                _, sc = orc.guppy_event_means(batch['arena'][o[i]:o[i + 1]], batch['calib'][i], 0,
                                              int(ev_blocks[i]), w['scale'], w['shift'])
                wiv, wc = orc.unsplit_scan(sc, 0, (int(w['seg_last'][3]) + 1) * 15,
                                           float(batch['calib'][i]['sampling_rate']))
                cand_mismatch += int(wc != cnt[i] or wiv.tolist() != iv[i, :min(wc, iv.shape[1])].tolist())
        cpu_s = time.perf_counter() - c0
        cpu = {'value': ns / cpu_s, 'unit': 'reads/s', 'cores': 1, 'kind': 'port',
               'sample': 'first {} reads of the same batch, same stages, oracle/libpxo.so '
                         '(C restatement, gcc -O2 AVX2), single thread, {:.1f} s'.format(ns, cpu_s)}
        got = res[:ns]
        same = [f for f in got.dtype.names if np.array_equal(got[f], want[f], equal_nan=True)]
        concordance = {
            'reads_compared': ns,
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

    line = {
        'metric': {'demux': 'reads/s (segment+barcode)', 'polya': 'reads/s (segment+barcode+polyA)',
                   'chimera': 'reads/s (segment+barcode+chimera filter)',
                   'full': 'reads/s (segment+barcode+polyA+chimera filter)',
                   'segment': 'reads/s (normalise+segment)'}[args.workload],
        'value': value, 'unit': 'reads/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 (LSTM MFMA) / f64 (Viterbi) / i16 in', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[{}]: {} reads/GPU x ~{} int16 samples, stages {}'.format(
                       {'demux': 2, 'polya': 3, 'chimera': 3, 'full': 3, 'segment': 1}[args.workload], args.reads, args.samples,
                       {'demux': 'a1-a13 (scaler LSTM + Viterbi + barcode LSTMs)',
                        'polya': 'a1-a17 (+ poly(A) events/DP)',
                        'chimera': 'a1-a13 + a18/a19 (Guppy block means + window scan)',
                        'full': 'a1-a19 (+ poly(A) + Guppy block means + window scan)',
                        'segment': 'a1,a5,a7,a8 (injected scaling)'}[args.workload]),
                   'reads_per_gpu': args.reads, 'samples_per_read': args.samples,
                   'parallelism': 'reads sharded x{}'.format(world), 'device': info['name'],
                   'arch': info['arch'], 'compute_units': info['compute_units']},
        'roofline': roofline, 'cpu_baseline': cpu, 'concordance': concordance, 'extra': extra,
    }
    os.write(json_fd, (json.dumps(line) + '\n').encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
