"""Multi-GPU sharding of a run: reads are independent units.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, gloo on
CPU for tests).  The data path has NO collective: rank r owns the contiguous
block of reads [floor(r*n/R), floor((r+1)*n/R)).  Only the end-of-batch
bookkeeping is exchanged: an all-gather of fixed-size label records
(LABEL_DTYPE, 16 B/read) and an all-reduce(sum) of the int64 count table
[label x barcode slot x status] that the summary writers consume
(reference: io.py:236-332 FinalSummaryTracker counts the same keys).
"""
import numpy as np

from . import native as N

LABEL_DTYPE = np.dtype([('read_index', '<i4'), ('status', 'i1'), ('label', 'i1'),
                        ('barcode', 'i1'), ('phred', 'u1'), ('score', '<f4'),
                        ('adapter_end', '<i4')])
assert LABEL_DTYPE.itemsize == 16

LABEL_NAMES = ('pass', 'fail', 'artifact')
N_BARCODE_SLOTS = 5     # undetermined + BC1..BC4 (commandline.py:137-159)


def shard_range(n_items, rank, world):
    """Contiguous block of rank `rank` among `world` ranks."""
    return (rank * n_items) // world, ((rank + 1) * n_items) // world


def shard_by_samples(lengths, world, cap=100000, fixed_cost=30000):
    """Balanced contiguous split when read lengths are skewed: cost of a read =
    min(len, scan limit) + a constant for the fixed-size recurrent nets."""
    cost = np.minimum(np.asarray(lengths, dtype=np.int64), cap) + fixed_cost
    csum = np.concatenate([[0], np.cumsum(cost)])
    bounds = [int(np.searchsorted(csum, csum[-1] * r / world)) for r in range(world)] + [len(cost)]
    bounds[0] = 0
    return [(bounds[r], max(bounds[r], bounds[r + 1])) for r in range(world)]


def final_label_records(results, first_index=0):
    """Label records from the FINAL result dicts of the facade (label, barcode, status as
    NanoporeRead.report() emits them): what the sinks count.  A read that never got a label
    counts as 'fail' with an undetermined barcode, like FinalSummaryTracker.feed_results
    (io.py:274-277)."""
    rec = np.zeros(len(results), dtype=LABEL_DTYPE)
    rec['read_index'] = first_index + np.arange(len(results), dtype=np.int32)
    rec['status'] = [N.STATUS_CODE[r['status']] for r in results]
    rec['label'] = [LABEL_NAMES.index(r.get('label', 'fail')) for r in results]
    rec['barcode'] = [-1 if r.get('barcode') is None else r['barcode'] for r in results]
    rec['phred'] = [r.get('barcode_score', 0) for r in results]
    rec['adapter_end'] = -1
    return rec


def final_label_records_from_table(table, rows, positions, loose, first_index=0):
    """final_label_records without the dicts: `rows` of a settled signal_loader.ReadTable
    at input `positions`, plus the `loose` (position, dict) outcomes of reads that never
    opened.  Record k describes input read first_index + k."""
    n = len(rows) + len(loose)
    rec = np.zeros(n, dtype=LABEL_DTYPE)
    rec['read_index'] = first_index + np.arange(n, dtype=np.int32)
    rec['adapter_end'] = -1
    idx, at = np.asarray(rows, dtype=np.int64), np.asarray(positions, dtype=np.int64)
    if len(idx):
        label = table.label[idx]
        rec['status'][at] = table.status[idx]
        rec['label'][at] = np.where(label < 0, LABEL_NAMES.index('fail'), label)
        rec['barcode'][at] = np.where(table.has_barcode[idx], table.barcode[idx], -1)
        rec['phred'][at] = np.where(table.has_barcode[idx], table.barcode_phred[idx], 0)
    for p, r in loose:
        rec['status'][p], rec['label'][p], rec['barcode'][p] = N.STATUS_CODE[r['status']], \
            LABEL_NAMES.index('fail'), -1
    return rec


def label_records(results, first_index=0, adapter_state=3):
    """Compact per-read label records from pxg_read_result rows: the NUMERIC-stage verdict
    only (0 pass / 1 fail), for callers that never build result dicts (bench.py).  A run
    that goes through the facade exchanges final_label_records instead: the base-space
    rules (not_basecalled, sequence_too_short, unsplit_read, ...) change status and label
    after the GPU pass (signal_analyzer.py:262-286)."""
    rec = np.zeros(len(results), dtype=LABEL_DTYPE)
    rec['read_index'] = first_index + np.arange(len(results), dtype=np.int32)
    rec['status'] = results['status']
    rec['label'] = np.where(results['status'] == 0, 0, 1)
    rec['barcode'] = np.where(results['bc_called'] == 1, results['bc_label'], -1)
    rec['phred'] = results['bc_phred']
    rec['score'] = results['bc_score']
    rec['adapter_end'] = results['seg_last'][:, adapter_state]
    return rec


def count_table(records):
    """int64 [3 labels, 5 barcode slots, n statuses] histogram."""
    tbl = np.zeros((len(LABEL_NAMES), N_BARCODE_SLOTS, len(N.STATUS_NAMES)), dtype=np.int64)
    np.add.at(tbl, (records['label'].astype(np.int64), records['barcode'].astype(np.int64) + 1,
                    records['status'].astype(np.int64)), 1)
    return tbl


def _device_for(dist):
    import torch
    return torch.device('cuda', torch.cuda.current_device()) \
        if dist.get_backend() == 'nccl' else torch.device('cpu')


def gather_labels_start(results, dist=None, first_index=0, sizes=None, force=False, records=None):
    """Launch the all-gather of the label records of every rank (ragged shards
    are padded to the largest shard) and return ``finish() -> records``.  The
    collective runs asynchronously on RCCL's stream, so a caller can start the
    next batch on the GPU and collect the labels afterwards.  dist=None or world
    1: local records only (unless `force`, which tests use to drive the
    collective on a single GPU).  `sizes`: the per-rank shard sizes when the
    caller already knows them (static sharding) -- saves the size exchange and
    its host synchronisation."""
    rec = label_records(results, first_index) if records is None else records
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return lambda: rec
    import torch
    world = dist.get_world_size()
    dev = _device_for(dist)
    if sizes is None:
        n_local = torch.tensor([len(rec)], dtype=torch.int64, device=dev)
        gathered = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(gathered, n_local)
        sizes = [int(s.item()) for s in gathered]
    elif len(sizes) != world or sizes[dist.get_rank()] != len(rec):
        raise ValueError('sizes must list every rank\'s shard size')
    nmax = max(sizes)
    buf = np.zeros(nmax, dtype=LABEL_DTYPE)
    buf[:len(rec)] = rec
    mine = torch.from_numpy(buf.view(np.uint8).reshape(nmax, LABEL_DTYPE.itemsize)).to(dev)
    out = torch.empty((world * nmax, LABEL_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    work = dist.all_gather_into_tensor(out, mine, async_op=True)

    def finish():
        work.wait()
        allrec = out.cpu().numpy().reshape(world, nmax * LABEL_DTYPE.itemsize)
        parts = [np.frombuffer(allrec[r].tobytes(), dtype=LABEL_DTYPE)[:sizes[r]]
                 for r in range(world)]
        return np.concatenate(parts)
    return finish


def gather_labels(results, dist=None, first_index=0, sizes=None, force=False, records=None):
    """Blocking form of gather_labels_start."""
    return gather_labels_start(results, dist, first_index, sizes, force, records)()


def reduce_counts(records, dist=None):
    """All-reduce(sum) of the local count table."""
    tbl = count_table(records)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return tbl
    import torch
    t = torch.from_numpy(tbl).to(_device_for(dist))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def agree_max(value, dist=None):
    """max of one small integer over the ranks (value itself without a process group).  The
    session driver uses it twice: the number of batch rounds every rank takes part in, and,
    once per round, the abort flag -- 0, or 1 + the rank that failed -- so that ONE rank's
    exception stops ALL ranks instead of leaving them waiting in the final collectives."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(value)
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=_device_for(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def bind_to_gpu_numa(device=0, pci_bus_id=None):
    """Put THIS process on the CPUs of the NUMA node its GPU hangs off, so that the loader
    thread, the status rules and -- by first touch -- every staging arena allocated from now on
    sit next to the PCIe root the copies go through.  Eight ranks staging 57 GB/s each read
    ~460 GB/s of host DRAM: across the socket interconnect that does not scale.  Returns what was
    done ({'numa_node', 'cpus', 'pci'}); a box that does not say (no sysfs entry, node -1, a
    container without the CPUs) is left alone: {'numa_node': None, ...}.  Call it before the
    staging memory is allocated (arrays that exist already: copy them afterwards)."""
    import os
    info = {'numa_node': None, 'cpus': None, 'pci': pci_bus_id}
    try:
        if pci_bus_id is None:
            pci_bus_id = info['pci'] = N.device_pci_bus_id(device)
        if not pci_bus_id or not hasattr(os, 'sched_setaffinity'):
            return info
        base = '/sys/bus/pci/devices/' + pci_bus_id
        with open(base + '/numa_node') as fh:
            node = int(fh.read().strip())
        if node < 0:
            return info
        with open('/sys/devices/system/node/node{}/cpulist'.format(node)) as fh:
            local = _cpulist(fh.read())
        mine = local & set(os.sched_getaffinity(0))
        if not mine:
            return info
        os.sched_setaffinity(0, mine)
        info.update(numa_node=node, cpus=len(mine))
    except (OSError, ValueError):
        pass
    return info
