"""GPU-shaped session driver: inputs -> shards -> fat double-buffered batches -> sinks.

The reference's ``ProcessingSession`` (pipeline.py:193-301) feeds 128-read batches to a
``ProcessPoolExecutor`` of CPU workers through an asyncio loop and writes whatever comes
back in completion order.  On GPUs the same job has a different shape:

  * one process per GPU (``torch.distributed``; RCCL on GPUs, gloo in CPU tests); rank r owns
    one contiguous, cost-balanced block of the run's reads (``distributed.shard_by_samples``)
    -- no data-path collective;
  * inside a rank the block is cut into a few FAT batches (thousands of reads: the LSTM
    kernels want >= 512 read tiles in flight).  A loader thread opens the reads of batch k+1
    and packs their int16 samples into a page-locked staging arena while batch k is on the
    GPU; ``pxg_batch_stage`` copies the arena on the copy stream under the running kernels
    and ``pxg_batch_swap`` makes it resident (include/pxg.h);
  * the facade turns the records of each batch into the reference's result dicts
    (signal_analyzer.py), which go to the sinks (sinks.py) in INPUT order, so the output
    files do not depend on the number of GPUs or on the batch size;
  * at the end the [label x barcode x status] count table is all-reduced and rank 0 prints
    the run summary (io.py:236-332) and stitches the per-rank part files together.

What it keeps from the reference: the (-1, message, traceback) fatality convention of
``process_batch``, the per-read error logging, and the early stop when nothing is
basecalled (pipeline.py:250-260).
"""
import logging
import os
import queue
import shutil
import threading
import time

import numpy as np

from . import distributed as D
from . import native
from . import sinks
from .signal_analyzer import SignalAnalyzer
from .signal_loader import ReadTable, summary_columns

__all__ = ['GpuSession', 'SessionAborted', 'enumerate_reads']


class SessionAborted(RuntimeError):
    """Another rank failed: this rank stopped with it (pipeline.py:207-213 semantics)."""


def enumerate_reads(config, bundle=None):
    """Every (filename, read_id) of the run, in a deterministic order: the bundle's own
    order, or a sorted recursive walk of inputdir (pipeline.py:303-337 without the inotify
    branch) with fast5_file.get_read_ids per file (single-read files: by the thousand in one native
    call, get_read_ids_many)."""
    from .fast5_file import get_read_ids_many
    if bundle is not None:
        d = bundle.d
        return [(str(f), str(r)) for f, r in zip(d['filename'], d['read_id'])], d.get('duration')
    files = []
    top = config['inputdir']
    for dirpath, dirnames, filenames in os.walk(top):
        dirnames[:] = sorted(n for n in dirnames if not n.startswith('.'))
        for name in sorted(filenames):
            if name.startswith('.') or not name.lower().endswith('.fast5'):
                continue
            files.append(os.path.relpath(os.path.join(dirpath, name), top))
    return get_read_ids_many(files, top), None


class _Staging:
    """One int16 staging arena.  The loader thread only allocates (reserve); page-locking
    happens on the session's thread right before the copy (settle), because every call into
    the split calls of a GPU context belong to one host thread (include/pxg.h)."""

    def __init__(self, ctx):
        self.ctx, self.buf, self.pinned, self.retired = ctx, np.empty(0, dtype=np.int16), False, []

    def reserve(self, n_samples):
        if n_samples > len(self.buf):
            if self.pinned:
                self.retired.append(self.buf)          # still registered: unpin in settle()
            self.buf, self.pinned = native.page_exclusive(int(n_samples * 1.1) + 4096, np.int16), False
        return self.buf

    def settle(self):
        for old in self.retired:
            self.ctx.unpin(old)
        self.retired = []
        if not self.pinned and len(self.buf):
            self.ctx.pin(self.buf)
            self.pinned = True

    def release(self):
        """Take every page lock off (works with a context that has been closed: NativeContext.unpin)."""
        bufs = self.retired + ([self.buf] if self.pinned else [])
        self.retired, self.pinned = [], False
        try:
            for b in bufs:
                self.ctx.unpin(b)
        finally:
            self.buf = np.empty(0, dtype=np.int16)


class GpuSession:

    def __init__(self, config, dist=None, batch_reads=8192, logger=None):
        self.config, self.dist = config, dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.batch_reads = int(batch_reads)
        self.logger = logger or logging.getLogger('poreplex')
        # this rank's process -- loader thread, status rules, and by first touch the bundle arena and
        # the staging arenas allocated from here on -- goes to the NUMA node of ITS GPU (N ranks staging
        # ~57 GB/s each must not cross the socket interconnect); config['numa_bind'] = False opts out
        self.numa = {'numa_node': None, 'cpus': None, 'pci': None}
        if config.get('numa_bind', True) and native.NativeContext.__module__ == native.__name__:
            device = int(config.get('device_id', os.environ.get('LOCAL_RANK', 0)))
            self.numa = D.bind_to_gpu_numa(device)
        self.analyzer = SignalAnalyzer(config, batchid=self.rank)
        self.ctx, self.loader = self.analyzer.ctx, self.analyzer.loader
        # the two staging arenas live as long as the session: page-locking them happens with the first batches,
        # taking the lock off again (hipHostUnregister: ~45 ms per 1.3 GB arena on the MI355X host) in close()
        self.stagings = None
        self.timing = {'load_s': 0.0, 'gpu_wait_s': 0.0, 'facade_s': 0.0, 'sink_s': 0.0,
                       'collect_s': 0.0, 'swap_run_s': 0.0, 'take_s': 0.0, 'stage_s': 0.0,
                       'fill_s': 0.0, 'load_ms': [],       # fill_s: until batch 0 computes and batch 1 is on its way
                       'load_phases_ms': [],
                       # the run outside the batch loop: before the loader starts, the loop itself, releasing the
                       # staging arenas / closing the sinks, the final collectives + stitching
                       'setup_s': 0.0, 'loop_s': 0.0, 'teardown_s': 0.0, 'finish_s': 0.0}               # per batch: FAST5 walk, signals, text (all threads' calls), wait for the prefetch, prepare

    # ---- loader thread: batch k+1 is opened and packed while batch k computes ---------
    def _produce(self, batches, slots, out, stop):
        from .fast5_file import TIMING as f5_timing
        ahead = None
        try:
            for k, reads in enumerate(batches):
                staging = slots.get()                 # a staging arena nobody is copying from
                if stop.is_set() or staging is None:  # the session is being torn down
                    return
                t0 = time.perf_counter()
                before = dict(f5_timing)
                if ahead is not None:
                    ahead.join()
                ahead = None
                if k + 1 < len(batches):              # the next batch's files are opened beside this one's decode
                    ahead = threading.Thread(target=self.loader.prefetch_files, args=(batches[k + 1],), daemon=True)
                    ahead.start()
                t_join = time.perf_counter()
                batch = self.analyzer.prepare(reads, ReadTable(), reserve=staging.reserve)
                t_prep = time.perf_counter()
                need = int(batch.table.n_raw[np.asarray(batch.entered, dtype=np.int64)].sum()) if batch.entered else 0
                rows, arena, offsets, calib = self.loader.pack(batch.table, staging, need)
                self.timing['load_s'] += time.perf_counter() - t0
                self.timing['load_ms'].append(round((time.perf_counter() - t0) * 1e3, 1))
                self.timing['load_phases_ms'].append([round((f5_timing[key] - before[key]) * 1e3, 1) for key in ('walk_s', 'signals_s', 'text_s')]
                                                     + [round((t_join - t0) * 1e3, 1), round((t_prep - t_join) * 1e3, 1)])
                out.put((batch, staging, rows, arena, offsets, calib))
            out.put(None)
        except BaseException as exc:                   # surfaces in run() on the main thread
            out.put(exc)

    def run(self, reads=None, lengths=None, presharded_at=None):
        """Process this rank's share of `reads` (default: everything enumerate_reads finds).
        `presharded_at`: `reads` is already this rank's block of the run and starts at that
        global read index (each rank opened its own input shard).  Returns the run summary
        dict on every rank; files are complete when it returns.

        Failure path (pipeline.py:207-213,250-265: a failed batch or an early stop ends the
        WHOLE run with one logged message): whatever goes wrong on this rank -- a loader
        exception, a PxgError, the early stop -- the loader thread is drained and joined, the
        staging arenas unpinned, the part files closed; with several ranks the abort flag is
        agreed on once per batch round, so every rank leaves the loop in the same round and
        raises SessionAborted instead of hanging in the final collectives."""
        cfg = self.config
        t_enter = t_loop = time.perf_counter()
        if reads is None:
            reads, lengths = enumerate_reads(cfg, self.loader.bundle)
        if presharded_at is not None:
            lo, hi = int(presharded_at), int(presharded_at) + len(reads)
            mine = reads
        else:
            if lengths is not None and len(lengths) == len(reads):
                lo, hi = D.shard_by_samples(lengths, self.world)[self.rank]
            else:
                lo, hi = D.shard_range(len(reads), self.rank, self.world)
            mine = reads[lo:hi]
        batches = [mine[a:a + self.batch_reads] for a in range(0, len(mine), self.batch_reads)]

        outdir = cfg['outputdir']
        os.makedirs(outdir, exist_ok=True)
        part = '.part{:04d}'.format(self.rank)
        # output names as the CLI derives them (commandline.py:137-159) unless the caller did
        labels, barcodes, layout = sinks.setup_output_name_mapping(cfg)
        labels = cfg.get('label_names', labels)
        barcodes = cfg.get('barcode_names', barcodes)
        layout = cfg.get('output_layout', layout)
        sink_cfg = dict(cfg, fast5_output=cfg.get('fast5_output', False))
        summary = sinks.SequencingSummaryWriter(sink_cfg, outdir, labels, barcodes, suffix=part,
                                                header=self.rank == 0)
        fastq = sinks.FASTQWriter(outdir, layout, suffix=part) if cfg.get('fastq_output') else None

        slots, ready, stop = queue.Queue(), queue.Queue(maxsize=2), threading.Event()
        if self.stagings is None:
            self.stagings = [_Staging(self.ctx), _Staging(self.ctx)]
        stagings = self.stagings
        thread = None
        records, status_seen = [], {}
        t_start = time.perf_counter()
        state = {'done': 0}

        def take():
            item = ready.get()
            if isinstance(item, BaseException):
                raise item
            return item

        # without poly(A), the chimera scan and the event dump no stage reads behind the segmentation's scan
        # limit: the tail of a longer read stays on the host (pxg_batch_stage_prefix)
        limit = {}
        if hasattr(self.ctx, 'prefix_limit_for'):
            whole = bool(self.loader.scan_unsplit or self.analyzer.dump_events)
            limit = {'prefix_limit': self.ctx.prefix_limit_for(self.loader.stage_mask, whole)}

        def stage(item):
            """Start the H2D copy of a packed batch into the spare input slot (copy stream)."""
            if item is not None and len(item[2]):
                item[1].settle()
                if isinstance(item[3], native.EncodedSamples):     # compressed bundle: bytes + chunk records
                    self.ctx.stage_z(item[3], item[4], item[5], **limit)
                else:
                    self.ctx.stage(item[3], item[4], item[5], **limit)
                return True
            return False

        def host_side(current):
            """Status rules, sinks and label records of one batch (under the next one's kernels)."""
            batch = current[0]
            t0 = time.perf_counter()
            table = batch.table
            if self.analyzer.dump_adapter or self.analyzer.dump_events:      # a dump file per batch, named by its first read
                self.analyzer.begin_dumps(lo + state['done'])
            self.analyzer.settle(batch)
            self.analyzer.flush_dumps()
            rows_in, positions, loose = batch.in_input_order()
            results = self.analyzer.finish(batch, input_order=True) if fastq is not None else None
            self.timing['facade_s'] += time.perf_counter() - t0
            t0 = time.perf_counter()
            idx = np.asarray(rows_in, dtype=np.int64)
            for code, n_seen in zip(*np.unique(table.status[idx], return_counts=True)):
                name = native.STATUS_NAMES[code]
                status_seen[name] = status_seen.get(name, 0) + int(n_seen)
            for _, r in loose:
                status_seen[r['status']] = status_seen.get(r['status'], 0) + 1
            for message in [table.error_message[i] for i in rows_in if table.error_message[i]] + \
                    [r['error_message'] for _, r in loose if 'error_message' in r]:
                self.logger.error(message)
            # the sinks, from the columns: only labelled reads have a summary row (io.py:166-168)
            labelled = idx[table.label[idx] >= 0]
            if not summary.write_table_rows(table, labelled):
                summary.write_columns(summary_columns(table, labelled, bool(cfg['barcoding']),
                                                      bool(cfg['measure_polya'])))
            if fastq is not None:
                fastq.write_sequences(results)
            records.append(D.final_label_records_from_table(table, rows_in, positions, loose,
                                                            first_index=lo + state['done']))
            state['done'] += len(rows_in) + len(loose)
            self.timing['sink_s'] += time.perf_counter() - t0
            self._check_early_stop(status_seen)

        failure = None
        try:
            self.loader.pin_bundle()      # batches of consecutive bundle reads are staged in place
            for s_ in stagings:
                slots.put(s_)
            # every rank takes part in the same number of rounds (one abort agreement per round)
            n_rounds = D.agree_max(len(batches), self.dist)
            thread = threading.Thread(target=self._produce, args=(batches, slots, ready, stop), daemon=True)
            thread.start()
            self.timing['setup_s'] = time.perf_counter() - t_enter
            t_loop = time.perf_counter()

            # pipeline: [k computes] [k+1 is copied under it] [k-1 is judged and written on the host].
            # The copy of k+1 is enqueued the moment the spare slot is free -- right after k became
            # resident -- so it runs under BOTH the kernels of k and the host work of k-1.
            current, nxt, nxt_staged, after, after_staged = None, None, False, None, False
            for rnd in range(max(n_rounds, 1)):
                try:
                    if failure is None and rnd == 0:
                        t_fill = time.perf_counter()
                        current = take()
                        if current is not None:
                            if stage(current):
                                self.ctx.swap()
                                self.ctx.run(self.loader.stage_mask)      # asynchronous: kernels are enqueued
                            slots.put(current[1])
                            nxt = take()
                            nxt_staged = stage(nxt)
                        self.timing['fill_s'] += time.perf_counter() - t_fill
                    if failure is None and current is not None:
                        t0 = time.perf_counter()
                        self.loader.collect_resident(current[0].table, current[2], current[4])   # D2H of the records: waits for run k
                        t1 = time.perf_counter()
                        self.timing['collect_s'] += t1 - t0
                        after, after_staged = None, False
                        if nxt is not None:
                            if nxt_staged:
                                self.ctx.swap()                       # k+1 resident (waits for its copy) ...
                                self.ctx.run(self.loader.stage_mask)  # ... and computing while k is written out
                            t2 = time.perf_counter()
                            self.timing['swap_run_s'] += t2 - t1
                            slots.put(nxt[1])
                            after = take()                            # k+2, packed by the loader thread meanwhile
                            t3 = time.perf_counter()
                            self.timing['take_s'] += t3 - t2
                            after_staged = stage(after)               # its copy starts now, under run k+1
                            self.timing['stage_s'] += time.perf_counter() - t3
                        self.timing['gpu_wait_s'] += time.perf_counter() - t0
                        host_side(current)                            # under the kernels of k+1
                        current, nxt, nxt_staged = nxt, after, after_staged
                except Exception as exc:      # noqa: BLE001 -- agreed on below, raised after the cleanup
                    failure = exc
                flag = D.agree_max(0 if failure is None else 1 + self.rank, self.dist)
                if flag:
                    if failure is None:
                        failure = SessionAborted('rank {} stopped the run (see its log)'.format(flag - 1))
                    break
        except BaseException as exc:          # KeyboardInterrupt and friends: clean up, do not agree
            failure = exc
        finally:
            t_down = time.perf_counter()
            self.timing['loop_s'] = t_down - t_loop
            # drain and join the loader thread: it may sit in slots.get() or ready.put()
            stop.set()
            if thread is not None:
                while thread.is_alive():
                    slots.put(None)
                    try:
                        while True:
                            ready.get_nowait()
                    except queue.Empty:
                        pass
                    thread.join(timeout=0.05)
            try:
                self.ctx.sync()               # nothing may still read a staging arena
            except Exception:                 # noqa: BLE001 -- the original failure is the one to report
                pass
            if failure is not None:           # a failed run leaves nothing page-locked behind
                self.close()
            summary.close()
            if fastq is not None:
                fastq.close()
            self.timing['teardown_s'] = time.perf_counter() - t_down
        if failure is not None:
            self.logger.error('Stopping the run: %s', failure)
            raise failure
        wall = time.perf_counter() - t_start
        t_finish = time.perf_counter()

        local = np.concatenate(records) if records else np.zeros(0, dtype=D.LABEL_DTYPE)
        counts = D.reduce_counts(local, self.dist)                    # all-reduce (RCCL / gloo)
        gathered = D.gather_labels(None, self.dist, records=local)    # all-gather of the labels
        if self.dist is not None:
            self.dist.barrier()                                       # every part file is closed
        out = {'reads': int(counts.sum()), 'reads_this_rank': int(len(local)), 'wall_s': wall,
               'rank': self.rank, 'world': self.world, 'batches': len(batches),
               'timing': dict(self.timing), 'labels': gathered, 'counts': counts, 'numa': self.numa}
        if self.rank == 0:
            self._stitch(outdir, 'sequencing_summary.txt')
            if fastq is not None:
                for name in set(layout.values()):
                    self._stitch(outdir, os.path.join('fastq', name + '.fastq.gz'))
            tracker = sinks.FinalSummaryTracker(labels, barcodes)
            tracker.feed_counts(counts, label_order=D.LABEL_NAMES)
            out['tracker'] = tracker
        if self.dist is not None:
            self.dist.barrier()
        out['timing']['finish_s'] = time.perf_counter() - t_finish
        return out

    def close(self):
        """Release the staging arenas (idempotent; also runs when the session object goes away)."""
        stagings, self.stagings = getattr(self, 'stagings', None), None
        for s_ in stagings or []:
            try:
                s_.release()
            except Exception:                 # noqa: BLE001 -- e.g. the context is gone already
                pass

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _check_early_stop(self, seen):
        """pipeline.py:250-260: stop when reads keep arriving without basecalls."""
        trigger = self.config.get('nobasecall_stop_trigger')
        if trigger and seen.get('okay', 0) == 0 and seen.get('not_basecalled', 0) >= trigger:
            raise RuntimeError(
                'Early stopping: {} out of {} reads are not basecalled. Please check if the files '
                "are correctly analyzed, or add `--basecall' to the command line.".format(
                    seen['not_basecalled'], sum(seen.values())))

    def _stitch(self, outdir, relname):
        """Rank 0: concatenate the per-rank part files in rank order (contiguous shards ->
        global input order).  gzip members concatenate into a valid gzip file."""
        final = os.path.join(outdir, relname)
        with open(final, 'wb') as dst:
            for r in range(self.world):
                part = '{}.part{:04d}'.format(final, r)
                if os.path.exists(part):
                    with open(part, 'rb') as src:
                        shutil.copyfileobj(src, dst)
                    os.unlink(part)
