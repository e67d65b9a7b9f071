"""N>1 path on CPU: 2 ranks over gloo (the GPU run uses the same code over
RCCL).  Reads shard contiguously; only label records and the count table are
exchanged."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from poreplex_amd import native as N
from poreplex_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 10000, 1000003):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_by_samples_balances_cost():
    rng = np.random.default_rng(0)
    lens = np.concatenate([rng.integers(9000, 20000, 500), rng.integers(200000, 900000, 500)])
    spans = D.shard_by_samples(lens, 4)
    assert spans[0][0] == 0 and spans[-1][1] == len(lens)
    cost = np.minimum(lens, 100000) + 30000
    per = [cost[a:b].sum() for a, b in spans]
    assert max(per) / min(per) < 1.15


def test_label_records_and_count_table():
    r = np.zeros(6, N.RESULT_DTYPE)
    r['status'] = [0, 0, 3, 5, 0, 4]
    r['bc_called'] = [1, 0, 0, 0, 1, 0]
    r['bc_label'] = [2, -1, -1, -1, 0, -1]
    r['seg_last'][:, 3] = [400, 380, -1, -1, 500, -1]
    rec = D.label_records(r, first_index=100)
    assert rec['read_index'].tolist() == list(range(100, 106))
    assert rec['barcode'].tolist() == [2, -1, -1, -1, 0, -1]
    tbl = D.count_table(rec)
    assert tbl.sum() == 6 and tbl[0, 3, 0] == 1 and tbl[0, 1, 0] == 1 and tbl[1, 0, 3] == 1


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch.distributed as dist
    from poreplex_amd import native as N
    from poreplex_amd import distributed as D
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    n_total = 1001
    a, b = D.shard_range(n_total, rank, world)
    # deterministic fake per-read records for this rank's shard
    res = np.zeros(b - a, N.RESULT_DTYPE)
    gi = np.arange(a, b)
    res['status'] = np.where(gi % 17 == 0, 3, 0)
    res['bc_called'] = (gi % 3 == 0) & (res['status'] == 0)
    res['bc_label'] = np.where(res['bc_called'] == 1, gi % 4, -1)
    res['bc_score'] = (gi % 100) / 100.0
    res['seg_last'][:, 3] = gi % 600
    allrec = D.gather_labels(res, dist, first_index=a)
    tbl = D.reduce_counts(D.label_records(res, a), dist)
    assert len(allrec) == n_total and allrec['read_index'].tolist() == list(range(n_total))
    g = allrec['read_index']
    assert np.array_equal(allrec['status'], np.where(g % 17 == 0, 3, 0))
    assert np.array_equal(allrec['barcode'], np.where((g % 3 == 0) & (g % 17 != 0), g % 4, -1))
    assert np.array_equal(tbl, D.count_table(allrec)) and tbl.sum() == n_total
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join({out!r}, 'rank%d.ok' % rank), 'w').write('ok')
''')


def test_two_rank_gloo_gather_and_reduce(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
        env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / 'rank0.ok').exists() and (tmp_path / 'rank1.ok').exists()


def test_bench_self_spawns_two_ranks_over_gloo():
    """`python bench.py --gpus 2` with no launcher in the environment: the script re-launches
    itself under torch.distributed.run, both ranks take part (counted by the collective),
    the label records of the whole run arrive on rank 0 with a GLOBAL, duplicate-free
    read_index, and stdout carries exactly one JSON line.  The GPU context is replaced by the
    oracle test double by tests/bench_standin.py (bench.py itself has no such seam; the line is marked TEST-STANDIN)."""
    import json
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['PYTHONPATH'] = os.path.join(ROOT, 'tests') + os.pathsep + env.get('PYTHONPATH', '')
    for scaling, extra, total in (('weak', ['--reads', '12', '--total-reads', '31', '--strong-base-reads', '5'], 24),
                                  ('strong', ['--total-reads', '25', '--base-reads', '7'], 25)):
        out = subprocess.run(
            [sys.executable, os.path.join(ROOT, 'tests', 'bench_standin.py'), '--gpus', '2', '--steps', '2',
             '--warmup', '1', '--samples', '12000', '--cpu-sample', '0', '--cpu-all-cores-sample', '0',
             '--scaling', scaling] + extra,
            env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, out.stdout
        line = json.loads(lines[0])
        assert line['n_gpus'] == 2 and line['scaling'] == scaling and line['value'] is None
        assert line['data'].startswith('TEST-STANDIN')
        assert line['extra']['ranks_counted_by_collective'] == 2
        assert line['extra']['labels_gathered'] == total
        assert line['extra']['labels_read_index_unique'] is True
        assert sum(line['config']['reads_per_gpu']) == total
        x = line['extra']
        if scaling == 'weak':
            # the legs the bare N > 1 call carries: every rank staged batches over "PCIe" at the same
            # time (min / mean / max over the ranks by collective), and BASELINE configs[4] -- one
            # seeded run sharded over the ranks -- ran beside the weak value
            for key in ('pcie_overlapped_reads_per_s', 'pcie_overlapped_encoded_reads_per_s', 'h2d_GBps'):
                st = x['pcie_over_ranks'][key]
                assert st['min'] <= st['mean'] <= st['max'] and st['min'] > 0, (key, st)
            c4 = x['configs4_strong']
            assert c4['scaling'] == 'strong' and c4['total_reads'] == 31 and sum(c4['reads_per_gpu']) == 31
            assert c4['labels_gathered'] == 31 and c4['labels_read_index_unique'] is True
            assert c4['value'] is None and c4['distinct_reads'] == 5
            assert 'numa' in x
        else:
            assert 'configs4_strong' not in x


def test_bench_end_to_end_two_ranks_over_gloo():
    """bench.py --end-to-end --gpus 2: each rank writes its shard as a read bundle, the session
    driver runs it (loader thread, batches, facade, sinks, all-gather / all-reduce); rank 0
    stitches one sequencing_summary.txt with a row per labelled read of BOTH shards."""
    import json
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['PYTHONPATH'] = os.path.join(ROOT, 'tests') + os.pathsep + env.get('PYTHONPATH', '')
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'tests', 'bench_standin.py'), '--gpus', '2', '--end-to-end', '--reads', '14',
         '--batch-reads', '5', '--samples', '12000', '--cpu-sample', '0', '--cpu-all-cores-sample', '0'],
        env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    x = line['extra']
    assert line['n_gpus'] == 2 and x['ranks_counted_by_collective'] == 2
    assert x['labels_gathered'] == 28 and x['labels_read_index_unique'] is True
    assert x['summary_rows'] == 28 and x['reads_labelled_pass'] >= 20


@pytest.mark.gpu
def test_bench_two_ranks_sharing_the_gpu_with_real_kernels():
    """The same bare `bench.py --gpus 2` call with the REAL context: both ranks on GPU 0
    (PXG_BENCH_SHARE_GPU=1, the bench's collectives over gloo).  Two processes drive the kernels
    at the same time, every rank's records are bit-identical to the oracle's (concordance block of
    rank 0), the labels of both ranks arrive once each, and configs[4] -- one seeded run sharded
    over the ranks -- runs beside the weak value.  A plumbing check: the line says so."""
    import json
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['PXG_BENCH_SHARE_GPU'] = '1'
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
         '--reads', '600', '--total-reads', '1500', '--strong-base-reads', '64', '--samples', '20000',
         '--cpu-sample', '16', '--cpu-all-cores-sample', '0'],
        env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    x = line['extra']
    assert line['n_gpus'] == 2 and line['value'] > 0 and 'SHARING one GPU' in line['data']
    assert x['ranks_counted_by_collective'] == 2 and x['labels_gathered'] == 1200 and x['labels_read_index_unique'] is True
    assert x['configs4_strong']['labels_gathered'] == 1500 and x['configs4_strong']['labels_read_index_unique'] is True
    assert sorted(x['configs4_strong']['reads_per_gpu']) == [750, 750]
    for key in ('pcie_overlapped_reads_per_s', 'pcie_overlapped_encoded_reads_per_s', 'h2d_GBps'):
        assert x['pcie_over_ranks'][key]['min'] > 0
    c = line['concordance']
    assert c is None or (c['status_mismatch'] == 0 and c['all_fields_bit_exact'])
    # BASELINE configs[4] as the headline of its own line: --scaling strong, ONE seeded run of 1 501 reads
    # over the two ranks -- an odd total, so the shards are uneven (751 + 750) --, each rank bound to the NUMA node
    # of its GPU, every label arriving once with its global read_index
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
         '--scaling', 'strong', '--total-reads', '1501', '--base-reads', '96', '--samples', '20000',
         '--cpu-sample', '0', '--cpu-all-cores-sample', '0'],
        env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][0])
    x = line['extra']
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['value'] > 0
    assert sorted(line['config']['reads_per_gpu']) == [750, 751] and line['config']['total_reads_per_step'] == 1501
    assert 'configs[4]' in line['config']['workload']
    assert x['ranks_counted_by_collective'] == 2 and x['labels_gathered'] == 1501 and x['labels_read_index_unique'] is True
    assert 'numa' in x and 'numa_node' in x['numa']
    # the session driver, two ranks, one GPU: loader threads, staging, sinks, label gather, count all-reduce
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--end-to-end', '--reads', '300',
         '--batch-reads', '100', '--samples', '20000', '--cpu-sample', '0', '--cpu-all-cores-sample', '0'],
        env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][0])
    x = line['extra']
    assert line['n_gpus'] == 2 and line['value'] > 0 and x['ranks_counted_by_collective'] == 2
    assert x['labels_gathered'] == 600 and x['labels_read_index_unique'] is True and x['summary_rows'] == 600
