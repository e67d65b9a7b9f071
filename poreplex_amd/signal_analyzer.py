"""Per-read processor: the reference's operator surface over the GPU hot path.

Drop-in for poreplex/signal_analyzer.py: ``process_batch(batchid, reads,
config)`` (picklable, :46-58), ``SignalAnalyzer(config, batchid)`` context
manager with ``.process(reads)`` (:61-134) and ``SignalAnalysis`` (:214-466).
Same signatures, result-dict schema, status/label vocabulary and result order.

How it differs inside: the reference walks reads one object at a time in
Python; here every numeric stage between int16 DAQ samples and the per-read
record runs in HIP behind the C ABI (include/pxg.h) in ONE pass per batch, and
what is left on the host is written over the columns of the batch's
``ReadTable`` (signal_loader.py) -- status / label rules are mask operations
on the ``pxg_read_result`` array, only the steps that must open a read's
basecall group run per read.  There is no CPU fallback: a missing library or
GPU surfaces as the ``(-1, msg, traceback)`` tuple the pipeline treats as fatal
(pipeline.py:207-213).
"""
import gc
import multiprocessing as mp
import os
import sys
import threading
import time
import traceback
from contextlib import AbstractContextManager
from hashlib import sha1
from weakref import proxy

import numpy as np

from . import native
from .fast5_file import decode_layout
from .signal_loader import LABELS, NanoporeRead, ReadTable, SignalAnalysisError
from .utils import union_intervals  # noqa: F401  (re-exported like the reference)
from .worker_persistence import WorkerPersistenceStorage

__all__ = ['SignalAnalyzer', 'SignalAnalysis', 'process_batch']

# Worker calls from many THREADS of one process (a thread pool in place of the reference's process pool, so that small
# calls share one context and one GPU batch): the two Python phases of a call -- open the reads into a table; status
# rules + result dicts -- are short (~0.1 ms each per 128 reads) but run under the interpreter lock, and a dozen
# threads contending for it turn each phase into milliseconds of waiting (measured: prepare 0.09 -> 5 ms, dicts
# 0.13 -> 1.4 ms at 32 threads, growing with the square of the thread count).  The phases therefore QUEUE on one plain
# lock: a thread waiting for it sleeps outside the interpreter lock instead of fighting for it, and the one thread that
# holds it runs its phase undisturbed.  The GPU pass in between holds neither.  (PXG_NO_HOST_PHASE_LOCK=1: off.)
_HOST_PHASE = threading.Lock()
_USE_HOST_PHASE = os.environ.get('PXG_NO_HOST_PHASE_LOCK') is None
# (A/B and tests: PXG_NO_PLAIN_RUN=1 sends every call through the batch table)
_PLAIN_RUN = os.environ.get('PXG_NO_PLAIN_RUN') is None
_BULK_UNSPLIT = os.environ.get('PXG_NO_BULK_UNSPLIT') is None      # (A/B and tests: candidates judged read by read)
_PLAIN_RUN_FAST5 = os.environ.get('PXG_NO_PLAIN_RUN_FAST5') is None    # (A/B: only bundle reads take the short path)
# reads too short for the scaler inside a plain run: True (they ride along), False (such a call takes the batch table:
# PXG_NO_SHORT_IN_RUN=1, the behaviour until then)
_SHORT_IN_RUN = os.environ.get('PXG_NO_SHORT_IN_RUN') is None
_FUSED_CALL = os.environ.get('PXG_NO_FUSED_CALL') is None         # (A/B: FAST5 decode and GPU pass as separate native calls)
_WORKER_IDS = {}            # process name -> the 16 hex digits dump files carry (signal_analyzer.py:163)
PLAIN_RUN_CALLS = 0         # worker calls that took SignalAnalyzer.process_plain_run (bench.py reports it)


class _NoLock:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


CALL_TRACE = None     # a list: (start, prepared, GPU pass done, dicts built) of every process() call

# the two message layouts downstream log parsers know (signal_analyzer.py:56-58,141-146)
BATCH_ERROR_FORMAT = '[{filename}:{lineno}] Unhandled exception {name}: {msg}'
READ_ERROR_FORMAT = ('[{srcfilename}:{lineno}] ({f5filename}#{read_id}) Unhandled '
                     'exception {name}: {msg}\n{exc}')


def _where_caught(excinfo):
    """(source file name, line) of the frame that caught the exception, plus the
    formatted traceback."""
    tb = excinfo[2]
    return (os.path.basename(tb.tb_frame.f_code.co_filename), tb.tb_lineno,
            ''.join(traceback.format_exception(*excinfo)))


def process_batch(batchid, reads, config):
    """Worker entry point (top level, so it pickles into a ProcessPoolExecutor).  Returns
    the result-dict list, or the fatal ``(-1, message, traceback)`` tuple."""
    analyzer = None
    try:
        analyzer = SignalAnalyzer(config, batchid)
        return analyzer.process(reads)
    except Exception as exc:
        srcname, lineno, text = _where_caught(sys.exc_info())
        return (-1, BATCH_ERROR_FORMAT.format(filename=srcname, lineno=lineno,
                                              name=type(exc).__name__, msg=str(exc)), text)
    finally:
        if analyzer is not None:
            analyzer.close()


class _Batch:
    """One worker batch between prepare() and finish(): its table, the outcomes that were
    settled before the GPU pass (finished dicts, or rows of reads that opened and stopped),
    the rows that entered the pass, and everybody's position in the input list."""

    def __init__(self, table):
        self.table = table
        self.early, self.early_at = [], []
        self.entered, self.entered_at = [], []
        self.judged = False

    def in_input_order(self):
        """(rows, positions, loose): table rows of the batch sorted by input position with
        those positions, and the outcomes that never became rows (file vanished / could not
        be opened) as (position, dict) pairs."""
        placed = [(p, r) for p, r in zip(self.early_at, self.early) if not isinstance(r, dict)]
        placed += list(zip(self.entered_at, self.entered))
        placed.sort()
        loose = [(p, r) for p, r in zip(self.early_at, self.early) if isinstance(r, dict)]
        return [r for _, r in placed], [p for p, _ in placed], loose

    def settle(self, position, outcome):
        self.early.append(outcome)
        self.early_at.append(position)

    def sort_by_position(self):
        """Both lists in input order (the bulk and the per-read admissions were appended apart)."""
        for name in ('early', 'entered'):
            at = getattr(self, name + '_at')
            if len(at) > 1 and (np.diff(np.asarray(at, dtype=np.int64)) < 0).any():
                order = sorted(range(len(at)), key=at.__getitem__)
                setattr(self, name, [getattr(self, name)[k] for k in order])
                setattr(self, name + '_at', [at[k] for k in order])


class SignalAnalyzer(AbstractContextManager):
    """Context manager; `with SignalAnalyzer(config, batchid) as analyzer` yields itself."""

    def __init__(self, config, batchid):
        WorkerPersistenceStorage(config).retrieve_objects(self)
        self.config = config
        self.inputdir, self.outputdir = config['inputdir'], config['outputdir']
        self.batchid, self.formatted_batchid = batchid, format(batchid, '08d')
        self.dump_adapter = bool(config.get('dump_adapter_signals'))
        self.loader.dump_adapter = self.dump_adapter
        self.dump_events = self.loader.dump_events = bool(config.get('dump_basecalls'))
        name = mp.current_process().name
        self.workerid = _WORKER_IDS.get(name) or _WORKER_IDS.setdefault(name, sha1(name.encode()).hexdigest()[:16])
        self.begin_dumps(batchid)
        self.loader.stage_mask = (
            native.STAGE_SCALER | native.STAGE_SEGMENT
            | (native.STAGE_BARCODE if config['barcoding'] else 0)
            | (native.STAGE_POLYA if config['measure_polya'] else 0))
        self.loader.scan_unsplit = bool(config.get('filter_unsplit_reads'))
        self.prebuilt = None        # (reads, their per-call FAST5 bundle) between process_plain_run and prepare
        self.call_arena = None      # that bundle's sample arena, the loader's to have back (process)

    # ---- batch driver --------------------------------------------------------
    def process(self, reads):
        """Result list of signal_analyzer.py:82-134: whatever was decided before the GPU
        pass first (encounter order), then every read that entered it (input order)."""
        phase = _HOST_PHASE if _USE_HOST_PHASE else _NoLock()
        try:
            if _PLAIN_RUN:
                results = self.process_plain_run(reads, phase)
                if results is not None:
                    return results
            t0 = time.perf_counter()
            with phase:
                batch = self.prepare(reads, ReadTable(len(reads)))    # a table of its own: calls may overlap (threads)
            t1 = time.perf_counter()
            self.loader.fit_scalers(batch.table)     # scaling parameters AND every other numeric stage
            t2 = time.perf_counter()
            with phase:
                results = self.finish(batch)
            if CALL_TRACE is not None:               # bench.py: where a worker call spends its time
                CALL_TRACE.append((t0, t1, t2, time.perf_counter()))
            return results
        finally:
            # the sample arena of a per-call FAST5 bundle (process_plain_run) goes back to the loader's pool once the
            # call is over, whichever path reported: nothing a call returns points into it
            arena, self.call_arena, self.prebuilt = self.call_arena, None, None
            if arena is not None:
                self.loader.call_arenas.give(arena)

    def process_plain_run(self, reads, phase):
        """process() for the usual worker call, without a batch table: `reads` is a run of consecutive reads of the
        read bundle -- or of multi-read FAST5 files, in file order: the per-call bundle the native reader makes of them
        (SignalLoader.fast5_run_plan) --, all of them long enough for the scaler and regular in their basecall summary
        (ReadBundle.plain_run_columns), and the configuration asks for nothing that walks every read (dumps, on-the-fly
        basecalling, the opt-in adapter trimming).  The samples go to the GPU as the bundle's own arena, the records
        (spike rows, scan candidates) come back, and csrc/pxg_pyreport.c report_run applies the status / label rules and
        builds the dicts in one pass: ~0.1 ms of Python for 128 reads where prepare + judge + report take 0.35 -- the
        interpreter lock is what bounds worker threads that feed the GPU in reference-sized calls.  Reads
        the chimera scan found candidates in are judged by the batch table's rules (finish_some_from_pass).
        None = not such a call: the general path takes it, and defines what this one must return
        (tests/test_plain_run.py); a per-call bundle that was built on the way is left in self.prebuilt for it."""
        loader, cfg = self.loader, self.config
        b = loader.bundle
        if self.dump_adapter or self.dump_events or cfg['albacore_onthefly'] \
                or (cfg['trim_adapter'] and cfg.get('trim_adapter_as_intended')) \
                or type(reads) is not list or not reads or type(reads[0]) is not tuple \
                or (loader.scan_unsplit and not hasattr(self.ctx, 'process_batch_ex')):
            return None
        from_files = b is None or not b.has_file(reads[0][0])
        if from_files and not _PLAIN_RUN_FAST5:
            return None
        fast = native.load_pyhost()
        if fast is None or not hasattr(fast, 'report_run'):
            return None
        t0 = time.perf_counter()
        batch = None
        if from_files:
            try:
                runs = loader.fast5_call_runs(reads)      # (may open a file: not under the phase lock)
            except Exception:             # noqa: BLE001  (an odd file: the general path says what is wrong with it, per read)
                runs = None
            if runs is None:
                return None
        with phase:
            n = len(reads)
            if from_files:
                # the per-call bundle: laid out and described from the files' cached metadata; nothing is decoded yet
                try:
                    batch = loader.fast5_run_plan(reads, runs)
                except Exception:         # noqa: BLE001
                    batch = None
                if batch is None:
                    return None
                self.call_arena, layout, plain, first = batch.arena, batch.layout, batch.plain, batch.first
                fits = plain is not None and bool(plain['regular'][first:first + n].all())
            else:
                plain = b.plain_run_columns(loader.scaler_cfg)
                if plain is None:
                    return None
                first = b.index.get(reads[0], -1)
                if first < 0 or b.keys[first:first + n] != reads or not plain['regular'][first:first + n].all():
                    return None
                if b.broken and any(key[0] in b.broken for key in reads):      # (files that exist but cannot be opened)
                    return None
                fits = True
            scan = sel = short = None
            if fits and _SHORT_IN_RUN is not None and not plain['ok'][first:first + n].all():
                # reads too short for the scaler among them (a few per cent of a real run: most 128-read calls have one):
                # they travel with the run -- the pass gives them up at its own gate --, are reported as what the host's
                # gate says, and come FIRST in the result list, as the reference returns what stopped before the pass
                if not _SHORT_IN_RUN:
                    fits = False
                else:
                    short = ~plain['long_enough'][first:first + n]
            if fits and loader.scan_unsplit:
                # the window scan over the same resident batch (signal_loader.fit_scalers); a Move table of another
                # k-mer size or a bundle with several block strides: the general path
                blocks = plain['frame_blocks'][first:first + n]
                kmer_ok = plain['kmer_ok'][first:first + n]
                if short is not None:                         # (a read that stops before the scaler is not scanned)
                    blocks, kmer_ok = np.where(short, 0, blocks), kmer_ok | short
                sel = blocks > 0
                if not kmer_ok.all() or (plain['frame_stride'] is None and sel.any()):
                    fits = False
                elif sel.any():
                    scan = (plain['frame_first'][first:first + n], blocks, plain['frame_stride'])
            if fits:
                o = plain['offsets'][first:first + n + 1]
                offsets, calib = o - o[0], plain['calib'][first:first + n]
                arena = b.samples_run(first, first + n) if batch is None else layout['arena'][:offsets[-1]]
                call = None
                if batch is not None and _FUSED_CALL and hasattr(fast, 'decode_and_run') and hasattr(self.ctx, 'batch_ex_call'):
                    call = self.ctx.batch_ex_call(n, loader.stage_mask, scan, bool(loader.stage_mask & native.STAGE_POLYA))
        none_enters = fits and batch is None and short is not None and bool(short.all())
        if none_enters:
            # nothing of this call reaches the scaler: no pass (the general path would not make one either; reads from
            # files are decoded all the same -- a read that cannot be is that read's 'unknown_error', short or not)
            got = {'records': np.zeros(n, dtype=native.RESULT_DTYPE), 'spikes': None}
            scan, t1 = None, time.perf_counter()
            t2 = t1
        elif batch is not None:
            # decode (+ the GPU pass, when this call makes it itself) without the interpreter lock; whoever declines from
            # here on has the files read already: self.prebuilt
            with loader.decoding() as threads:
                if fits and call is not None:
                    t1 = time.perf_counter()
                    decoded, rc = loader.decode_and_run(fast, layout, threads, call, offsets, calib)
                    t2 = time.perf_counter()
                else:
                    decode_layout(layout, threads)
                    decoded = not (layout['signal_status'].any() or layout['basecall_status'].any())
            self.prebuilt = (reads, batch)
            if not (fits and decoded):
                return None
        elif not fits:
            return None
        else:
            loader.pin_bundle()
        if none_enters:
            pass
        elif batch is None or call is None:
            t1 = time.perf_counter()
            got = loader.records_of_run(arena, offsets, calib, scan)
            t2 = time.perf_counter()
        else:
            got = call.result(rc)
            if got is None:          # (a variable-size output outgrew its buffer: the wrapper's loop sizes it)
                got = loader.records_of_run(arena, offsets, calib, scan)
                t2 = time.perf_counter()
        with phase:
            some = np.nonzero(got['unsplit'][1])[0] if scan is not None else ()
            if len(some) > n // 4:
                # in-read adapter candidates (or failed scans) all over the call: the batch table, with the pass that
                # has already run
                if short is not None:
                    return None           # (... which has records of reads the table would not have sent: the general path)
                results = self.finish_from_pass(reads, got, sel)
            else:
                skip = None
                if len(some):            # a few reads with candidates: a table of just those, the rest in the C pass
                    skip = np.zeros(n, dtype=np.bool_)
                    skip[some] = True
                spikes = got.get('spikes')
                was_on = gc.isenabled()      # (nothing report_run builds can be part of a cycle: ReadTable.report)
                gc.disable()
                try:
                    results = fast.report_run(plain, first, n, got['records'], self.ctx.state_names.index('adapter'),
                                              bool(cfg['barcoding']), int(cfg['minimum_sequence_length']),
                                              tuple(native.STATUS_NAMES), tuple(LABELS), bool(cfg['measure_polya']),
                                              None if spikes is None else np.ascontiguousarray(spikes[0], dtype=np.float32),
                                              None if spikes is None else np.ascontiguousarray(spikes[1], dtype=np.int64),
                                              skip, None if short is None else np.ascontiguousarray(short))
                    if skip is not None:
                        held, held_first = (b, first) if batch is None else (batch.bundle(), 0)
                        for at, report in zip(some.tolist(), self.finish_some_from_pass(held, held_first, some, got)):
                            results[at] = report
                except (IndexError, TypeError, KeyError, ValueError):
                    if short is not None:
                        return None
                    results = self.finish_from_pass(reads, got, sel)      # columns it cannot read as they are
                finally:
                    if was_on:
                        gc.enable()
                if short is not None:     # what stopped before the pass first (encounter order), then the rest (input order)
                    flags = short.tolist()
                    results = [r for r, s_ in zip(results, flags) if s_] + [r for r, s_ in zip(results, flags) if not s_]
            global PLAIN_RUN_CALLS
            PLAIN_RUN_CALLS += 1
        if CALL_TRACE is not None:
            CALL_TRACE.append((t0, t1, t2, time.perf_counter()))
        return results

    def finish_from_pass(self, reads, got, scanned):
        """The batch table for a plain run whose GPU pass has been made (process_plain_run): open the reads, attach the
        records / spike rows / scan candidates as fit_scalers would have, judge and report."""
        loader = self.loader
        batch = self.prepare(reads, ReadTable(len(reads)))
        table = batch.table
        rows = loader.pack(table)[0]                 # (a run of bundle reads: nothing is copied)
        if len(rows) != len(got['records']):
            raise RuntimeError('plain run: {} reads entered the pass, {} records came back'.format(len(rows), len(got['records'])))
        loader.attach_records(table, rows, got['records'], got.get('spikes'))
        if scanned is not None and got.get('unsplit') is not None:
            loader.attach_unsplit(table, rows, scanned, got['unsplit'])
        return self.finish(batch)

    def finish_some_from_pass(self, bundle, first, some, got):
        """Result dicts of the reads `some` (positions in a plain run that starts at bundle read `first`) through a batch
        table of their own: the reads the chimera scan found candidates in, judged by the table's rules over the
        records, spike rows and candidate lists of the pass the whole run has made (process_plain_run)."""
        loader = self.loader
        table = ReadTable(len(some))
        rows = table.extend_from_bundle(bundle, first + some)
        table.pending[rows] = False                  # (the samples have been to the GPU)
        loader.attach_records(table, rows, got['records'], got.get('spikes'), gpu_rows=some)
        intervals, count, start = got['unsplit']
        table.unsplit_count[rows] = count[some]
        for k, at in enumerate(some.tolist()):
            if count[at] > 0:
                table.unsplit[rows[k]] = intervals[start[at]:start[at + 1]].tolist()
        self.judge(rows, table)
        table.release(rows)
        return table.report(rows)

    def prepare(self, reads, table=None, reserve=None):
        """Host-only first phase: open every read into a batch table.  The session driver
        runs this for batch k+1 while batch k is on the GPU (`reserve`: its staging arena, so
        FAST5 signals are decoded straight into page-locked memory)."""
        loader = self.loader
        table = loader.table if table is None else table
        batch = _Batch(table)
        prebuilt, self.prebuilt = self.prebuilt, None
        prebuilt = prebuilt[1] if prebuilt is not None and prebuilt[0] is reads else None
        reads = [tuple(r) for r in reads]
        where = loader.prepare_many(reads, table, reserve, prebuilt)   # bundle / FAST5 reads: column appends
        bulk = np.nonzero(where >= 0)[0]
        stopped = table.stopped[where[bulk]]
        for position in np.nonzero(where < 0)[0].tolist():     # everything else, read by read
            f5file, read_id = reads[position]
            if not loader.exists(f5file):
                batch.settle(position, {'filename': f5file, 'status': 'disappeared'})
                continue
            try:
                row = loader.prepare_loading(f5file, read_id, table).row
            except Exception as exc:
                batch.settle(position, self.pack_unhandled_exception(f5file, read_id, exc, sys.exc_info()))
                continue
            if table.stopped[row]:
                batch.settle(position, row)
            else:
                batch.entered.append(row)
                batch.entered_at.append(position)
        batch.early += where[bulk[stopped]].tolist()
        batch.early_at += bulk[stopped].tolist()
        batch.entered += where[bulk[~stopped]].tolist()
        batch.entered_at += bulk[~stopped].tolist()
        batch.sort_by_position()
        return batch

    def settle(self, batch):
        """Last phase without the dicts: status / label rules over the GPU records; the
        outcome of every read is then in the batch table's columns (batch.in_input_order())."""
        if not batch.judged:
            self.judge(batch.entered, batch.table)
            batch.table.release(batch.entered)
            batch.judged = True
        return batch

    def finish(self, batch, input_order=False):
        """Last phase: status / label rules over the GPU records, then the result dicts --
        early outcomes first as the reference returns them, or in input order."""
        self.settle(batch)
        early_rows = [r for r in batch.early if not isinstance(r, dict)]
        reports = iter(batch.table.report(early_rows + batch.entered))
        results = [r if isinstance(r, dict) else next(reports) for r in batch.early] + list(reports)
        if input_order:
            at = np.argsort(np.array(batch.early_at + batch.entered_at, dtype=np.int64), kind='stable')
            results = [results[i] for i in at]
        return results

    def judge(self, rows, table=None):
        """Status / label rules of SignalAnalysis.process (:230-286) for many rows."""
        t, cfg = (self.loader.table if table is None else table), self.config
        rows = t.live_rows(rows)
        if not len(rows):
            return
        rec = t.records[t.gpu_row[rows]]
        adapter = self.ctx.state_names.index('adapter')
        found = rec['seg_first'][:, adapter] >= 0
        t.halt(rows[~found], 'adapter_not_detected', 'fail')
        rows, rec = rows[found], rec[found]
        if self.dump_adapter:            # before anything later can fail (:243-244)
            self.queue_adapter_dumps(t, rows, rec)

        # barcodes are decided before anything base-space can fail, so reads that fail
        # later keep theirs (:243-244 queues the window first)
        if cfg['barcoding']:
            self.demuxer.assign(t, rows, rec)
        broken = np.zeros(len(rows), dtype=bool)
        if cfg['measure_polya']:
            for k in self.polyaanalyzer.assign(t, rows, rec):      # bulk; the odd ones per read
                broken[k] = not self._guarded(t, rows[k], self.polyaanalyzer, NanoporeRead(t, rows[k]))
        if self.dump_events:             # load_events + write_basecalled_events (:259-263), read by read
            for k in np.nonzero(~broken)[0].tolist():
                broken[k] = not self._guarded(t, rows[k], self.queue_event_dump, t, rows[k], rec[k])
        settled = self.bulk_base_space(t, rows, ~broken)
        if _BULK_UNSPLIT and cfg['filter_unsplit_reads']:
            settled |= self.bulk_unsplit_rule(t, rows, rec, ~broken & ~settled)
        for k in np.nonzero(~broken & ~settled)[0].tolist():
            broken[k] = not self._guarded(t, rows[k], self.base_space_checks, t, rows[k], rec[k])
        done = rows[~broken & ~t.stopped[rows]]
        t.label[done] = 0                # 'pass'

    def bulk_base_space(self, t, rows, todo):
        """base_space_checks as column operations, for rows whose basecall summary sits in a
        read bundle's columns (fast5_file.BASECALL_COLUMNS).  Returns the mask of rows it
        settled (passed, or halted with a domain status); everything irregular -- a missing
        or odd event table, a frame that does not fit the raw signal, reads the chimera scan
        flagged -- is left to the per-read path, which raises exactly as before."""
        cfg = self.config
        settled = np.zeros(len(rows), dtype=bool)
        if t.bundle is None or cfg['albacore_onthefly'] or (cfg['trim_adapter'] and cfg.get('trim_adapter_as_intended')):
            return settled      # (the opt-in trimming walks every read's event table: the per-read path)
        bi = t.bundle_index[rows]
        pick = np.nonzero(todo & (bi >= 0))[0]
        if not len(pick):
            return settled
        d, r, b = t.bundle.d, rows[pick], bi[pick]
        absent = ~d['bc_present'][b]
        t.halt(r[absent], 'not_basecalled', 'fail')
        settled[pick[absent]] = True
        # regular = Guppy frame that fits the raw signal (convert_events_guppy's size rule)
        first, stride, n_moves = d['bc_first_sample'][b], d['bc_block_stride'][b].astype(np.int64), d['bc_n_moves'][b]
        covered = np.maximum(np.minimum(first + stride * n_moves, t.n_raw[r]) - first, 0)
        kind = d['bc_table'][b]
        regular = ~absent & ((kind == 1) | (kind == 2)) & (n_moves >= 0) & (stride > 0) & \
            (-(-covered // np.maximum(stride, 1)) == n_moves) & t.has_scaling[r]
        if cfg['filter_unsplit_reads']:
            # the decision rule needs the event frame only for reads with candidates; the
            # frame's own validity (k-mer size of a Move table) is checked for everyone
            kmer = (d['seq_offsets'][b + 1] - d['seq_offsets'][b]) - d['bc_move_sum'][b] + 1
            regular &= (t.unsplit_count[r] == 0) & ((kind == 2) | (kmer == 5) | (kmer == 1))
        ok, okb = r[regular], b[regular]
        t.sequence_length[ok], t.mean_qscore[ok] = d['bc_sequence_length'][okb], \
            d['bc_mean_qscore'][okb].astype(np.float32)
        t.num_events[ok], t.has_summary[ok] = d['bc_num_events'][okb], True
        so = d['seq_offsets']
        t.seq_lazy[ok] = True             # the (sequence, qstring, 0) tuples are made on demand
        short = (so[okb + 1] - so[okb]) < cfg['minimum_sequence_length']      # trimming is a no-op
        t.halt(ok[short], 'sequence_too_short', 'fail')
        settled[pick[regular]] = True
        return settled

    def bulk_unsplit_rule(self, t, rows, rec, todo):
        """base_space_checks for the rows bulk_base_space left because the chimera scan found candidates in them: bundle
        reads whose basecall is a Guppy MOVE table with a regular frame and 5-mer or 1-mer states.  The reference's rule
        (signal_analyzer.py:420-443; SignalAnalysis.detect_unsplit_read is its restatement and stays the definition,
        tests/test_plain_run.py holds the two against each other) counts, per stretch between candidates, the bases
        whose best p_model_state among their events clears a limit.  For a Move table that probability is a function of
        the base's quality character alone (event_frame: 1 - 10^(-phred / 10) of the base the event sits on), the events
        start `block_stride` apart, and a base's events are the run between two moves -- so a stretch's count is a
        difference of prefix sums over the read's Move column and its event range comes from two divisions, without the
        event table.  Returns the mask of the rows settled here; Events tables, failed scans and
        everything irregular stay with the per-read path."""
        cfg = self.config
        settled = np.zeros(len(rows), dtype=bool)
        b = t.bundle
        if b is None or cfg['albacore_onthefly'] or (cfg['trim_adapter'] and cfg.get('trim_adapter_as_intended')) \
                or 'move_arena' not in b.d:
            return settled
        d, bi = b.d, t.bundle_index[rows]
        pick = np.nonzero(todo & (bi >= 0) & (t.unsplit_count[rows] > 0))[0]
        if not len(pick):
            return settled
        r, x = rows[pick], bi[pick]
        first, stride, n_moves = d['bc_first_sample'][x], d['bc_block_stride'][x].astype(np.int64), d['bc_n_moves'][x]
        covered = np.maximum(np.minimum(first + stride * n_moves, t.n_raw[r]) - first, 0)
        kmer = (d['seq_offsets'][x + 1] - d['seq_offsets'][x]) - d['bc_move_sum'][x] + 1
        fits = d['bc_present'][x].astype(bool) & (d['bc_table'][x] == 1) & (n_moves > 0) & (stride > 0) & \
            (-(-covered // np.maximum(stride, 1)) == n_moves) & t.has_scaling[r] & ((kmer == 5) | (kmer == 1)) & \
            (d['move_offsets'][x + 1] - d['move_offsets'][x] == n_moves)
        if not fits.any():
            return settled
        limits = cfg['unsplit_read_detection']
        # quality character -> does its base clear the limit: the expression of event_frame over every byte value
        phred = np.arange(256, dtype=np.uint8) - 33
        good_char = (1 - 10 ** -(phred / 10)) > np.float64(limits['basecount_quality_limit'])
        adapter = self.ctx.state_names.index('adapter')
        elspan = cfg['signal_processing']['rough_signal_stride']
        mo, so, moves, quals = d['move_offsets'], d['seq_offsets'], d['move_arena'], d['qual_arena']
        unsplit = np.zeros(len(pick), dtype=bool)
        for k in np.nonzero(fits)[0].tolist():
            row, i = int(r[k]), int(x[k])
            f0, st, n = int(first[k]), int(stride[k]), int(n_moves[k])
            mv = moves[mo[i]:mo[i + 1]]
            lead = 2 if kmer[k] == 5 else 0                              # (event_frame: the k-mer's leading half)
            if lead == 0 and mv[0] == 0:
                fits[k] = False           # an event before the first base of a 1-mer table: the per-read path's error
                continue
            pos = np.cumsum(mv, dtype=np.int64)
            good = good_char[quals[so[i]:so[i + 1]]][pos - 1 + lead]     # event -> the quality of the base it sits on
            changes = np.cumsum(good & (mv != 0))                        # bases that start at or before an event, good ones
            merged = union_intervals(t.unsplit[row])
            begins = [(int(rec['seg_last'][pick[k], adapter]) + 1) * elspan] + [iv[1] for iv in merged]
            ends = [iv[0] for iv in merged] + [None]
            hq = []
            for lo_at, hi_at in zip(begins, ends):
                # events whose start lies in [lo_at, hi_at]: start of event j = f0 + st * j
                lo = min(max(-(-(lo_at - f0) // st), 0), n)
                hi = n if hi_at is None else min(max((hi_at - f0) // st + 1, 0), n)
                hq.append(0 if hi <= lo else int(good[lo]) + int(changes[hi - 1] - changes[lo]))
            later = sum(hq[1:])
            unsplit[k] = later > limits['subread_basecount_limit'] or \
                (later + 1) / (hq[0] + 1) > limits['subread_baseratio_limit']
        # the summary load_fast5_events stores before the rule runs, then the rule's verdict, then the length rule
        ok, okb = r[fits], x[fits]
        t.sequence_length[ok], t.mean_qscore[ok] = d['bc_sequence_length'][okb], d['bc_mean_qscore'][okb].astype(np.float32)
        t.num_events[ok], t.has_summary[ok] = d['bc_num_events'][okb], True
        t.seq_lazy[ok] = True
        t.halt(r[fits & unsplit], 'unsplit_read', 'artifact')
        rest = fits & ~unsplit
        short = (so[x[rest] + 1] - so[x[rest]]) < cfg['minimum_sequence_length']
        t.halt(r[rest][short], 'sequence_too_short', 'fail')
        settled[pick[fits]] = True
        return settled

    def _guarded(self, t, row, fn, *args):
        """Run one read's step; a domain failure halts the read with its label, anything
        else becomes that read's 'unknown_error' (:118-122).  False if the read is out."""
        try:
            fn(*args)
            return not t.stopped[row]
        except SignalAnalysisError as exc:
            t.halt(row, exc.args[0], 'artifact' if exc.args[0] == 'unsplit_read' else 'fail')
        except Exception as exc:
            err = self.pack_unhandled_exception(t.filename[row], t.read_id[row], exc, sys.exc_info())
            NanoporeRead(t, row).set_error(err['status'], err['error_message'])
        return False

    def base_space_checks(self, t, row, record):
        """The part of :262-279 that needs the read's basecall group."""
        cfg = self.config
        analysis = SignalAnalysis(NanoporeRead(t, row), self)
        segments = analysis.segments_of(record)
        stride = cfg['signal_processing']['rough_signal_stride']
        events = analysis.load_events()
        if cfg['trim_adapter']:
            analysis.trim_adapter(events, segments, stride)
        if cfg['filter_unsplit_reads'] and analysis.detect_unsplit_read(events, segments, stride):
            raise SignalAnalysisError('unsplit_read')
        seq = t.sequence_of(row)
        if seq is not None and len(seq[0]) - seq[2] < cfg['minimum_sequence_length']:
            raise SignalAnalysisError('sequence_too_short')

    def pack_unhandled_exception(self, f5filename, read_id, exc, excinfo):
        srcname, lineno, text = _where_caught(excinfo)
        return {'filename': f5filename, 'read_id': read_id, 'status': 'unknown_error',
                'error_message': READ_ERROR_FORMAT.format(
                    srcfilename=srcname, lineno=lineno, f5filename=f5filename, read_id=read_id,
                    name=type(exc).__name__, msg=str(exc), exc=text)}

    def __exit__(self, exc_type, exc_value, tb):
        self.close()

    def close(self):
        """Flushes the dump file of the batch; the GPU context outlives it (worker_persistence)."""
        self.flush_dumps()

    # ---- --dump-adapter-signals (signal_analyzer.py:155-211,450-466) --------------------------
    def begin_dumps(self, batchid):
        """Start collecting the dumps of batch `batchid` (the session driver: once per batch)."""
        self.formatted_dump_batchid = format(batchid, '08d')
        self.adapter_dump_list, self.adapter_dump_seen = [], set()
        self.event_dump_list, self.event_dump_seen = [], set()

    def queue_adapter_dumps(self, t, rows, rec):
        """dump_adapter_signal for the rows whose adapter was found: the pooled + scaled signal
        of the adapter stretch (downloaded with the records, ReadTable.adapter_dump) and its
        catalogue row (read id, first raw sample, raw sample behind the last).  A read id that
        was dumped before in this batch keeps its first dataset and gets no second row."""
        values, offsets = t.adapter_dump
        adapter = self.ctx.state_names.index('adapter')
        stride = int(self.loader.scaler_cfg['stride'])
        first, last = rec['seg_first'][:, adapter], rec['seg_last'][:, adapter]
        g = t.gpu_row[rows]
        for k in np.nonzero(offsets[g + 1] > offsets[g])[0].tolist():
            read_id = t.read_id[rows[k]]
            if read_id in self.adapter_dump_seen:
                continue
            self.adapter_dump_seen.add(read_id)
            self.adapter_dump_list.append((read_id, values[offsets[g[k]]:offsets[g[k] + 1]],
                                           int(first[k]) * stride, (int(last[k]) + 1) * stride))

    def flush_dumps(self):
        """adapter-dumps/part-<worker>-<batch>.h5 with the reference's groups: the signals under
        adapter/<batch>/<read id> (float32), the catalogue under catalog/adapter/<batch>.  The
        reference appends every batch of a worker to ONE part-<worker>.h5; files are written whole
        here, so a batch gets its own -- the inventory builder (io.py:351-366) takes
        `part-*.h5` and whatever batch groups it finds in them."""
        self.flush_event_dumps()
        if not self.dump_adapter or self.adapter_dump_list is None:
            return
        from .fast5_write import H5Writer
        batch = self.formatted_dump_batchid
        top = os.path.join(self.outputdir, 'adapter-dumps')
        os.makedirs(top, exist_ok=True)
        catalog = np.array([(r.encode('ascii'), a, b) for r, _, a, b in self.adapter_dump_list],
                           dtype=[('read_id', 'S36'), ('start', 'i8'), ('end', 'i8')])
        with H5Writer(os.path.join(top, 'part-{}-{}.h5'.format(self.workerid, batch))) as h5:
            h5.require_group('adapter/' + batch)
            for read_id, values, _, _ in self.adapter_dump_list:
                h5.create_dataset('adapter/{}/{}'.format(batch, read_id), np.asarray(values, dtype=np.float32))
            h5.create_dataset('catalog/adapter/' + batch, catalog)
        self.adapter_dump_list = None


    # ---- --dump-basecalls (signal_analyzer.py:63-69,156,165-197,259-263,288-309) -------------
    EVENT_DUMP_FIELDS = [('mean', '<f4'), ('start', '<u8'), ('stdv', '<f4'), ('length', '<u8'),
                         ('model_state', 'S5'), ('move', '<i4'), ('pos', '<u8'), ('end', '<u8'),
                         ('scaled_mean', '<f8')]

    def queue_event_dump(self, t, row, record):
        """The event table load_events builds for one read (fast5_file.py:183-230,
        signal_analyzer.py:311-326) and the attributes of get_dump_attributes.  The base-space
        columns are made here from the basecall; mean, stdv and scaled_mean were computed on the
        resident batch (ReadTable.event_dump).  Raises what load_events raises."""
        read = NanoporeRead(t, row)
        bcall = read.load_fast5_events()
        fields = list(self.EVENT_DUMP_FIELDS)
        fields[4] = ('model_state', 'S{}'.format(self.kmersize))
        kind = bcall.get('table', 'move')
        if kind == 'albacore':
            # the table as the file holds it (fast5_file.py:178-179) + the three columns load_events adds
            # (:318-324); scaled_mean = np.poly1d(float32[scale, shift])(mean) in the width of `mean'
            frame = albacore_frame(bcall)
            ev = frame['table']
            for name in ('stdv', 'length', 'model_state'):
                if name not in ev:
                    raise KeyError(name)
            n_rows = len(frame['start'])
            table = np.zeros(n_rows, dtype=fields)
            mean = np.asarray(ev['mean'])
            scale, shift = (mean.dtype.type(v) for v in t.scale_shift[row]) if mean.dtype == np.float32 else \
                (np.float64(np.float32(v)) for v in t.scale_shift[row])
            scaled = scale * mean
            scaled = scaled + shift
            with np.errstate(all='ignore'):
                table['mean'], table['stdv'], table['scaled_mean'] = mean, ev['stdv'], scaled
                table['start'], table['length'], table['model_state'] = ev['start'], ev['length'], ev['model_state']
                table['move'], table['pos'], table['end'] = frame['move'], frame['pos'], frame['end']
        else:
            first, n_blocks, stride = read.guppy_event_geometry(bcall=bcall)
            g = int(t.gpu_row[row])
            if tuple(t.event_frame[g]) != (first, n_blocks, stride):
                raise Exception('event frame of the dump does not match the basecall')
            mean, stdv, scaled, offsets = t.event_dump[stride]
            at = slice(int(offsets[g]), int(offsets[g + 1]))
            moves = np.asarray(bcall['move'], dtype=np.int64)
            if kind == 'guppy_events':               # Events table of Guppy < 2.3.7: its own model_state column
                ev = bcall.get('events') or {}
                if 'model_state' not in ev:
                    raise KeyError('model_state')
                kmers = np.asarray(ev['model_state'])
            else:
                seq = bcall['sequence']
                kmer_size = len(seq) - int(moves.sum()) + 1
                rev = seq[::-1].replace('U', 'T')
                if kmer_size == 1:                       # flip-flop models: 1-mer frames shown as 5-mers
                    rev = '__' + rev + '__'
                elif kmer_size != 5:
                    raise Exception('Move table is encoded with an unknown kmer-size.')
                # revseq[pos : pos + 5] for pos = cumsum(move) - 1, as Python slices it (short or empty
                # at and beyond the end, and for a table that starts with a stay)
                text = np.frombuffer(rev.encode('ascii') + b'\0' * 5, dtype=np.uint8)
                pos0 = np.cumsum(moves) - 1
                pos0 = np.where((pos0 < 0) | (pos0 > len(rev)), len(rev), pos0)
                kmers = np.ascontiguousarray(np.lib.stride_tricks.sliding_window_view(text, 5)[pos0]).view('S5').ravel()
            table = np.zeros(n_blocks, dtype=fields)
            start = first + stride * np.arange(n_blocks, dtype=np.int64)
            table['mean'], table['stdv'], table['scaled_mean'] = mean[at], stdv[at], scaled[at]
            table['start'], table['length'], table['model_state'] = start, stride, kmers
            table['move'], table['pos'] = moves, np.cumsum(moves)
            table['end'] = np.append(start[1:], start[-1:] + 1) if n_blocks else start
        # get_dump_attributes (:288-309)
        adapter = self.ctx.state_names.index('adapter')
        pool = int(self.loader.scaler_cfg['stride'])
        attrs = [('signal_scale', np.float32(t.scale_shift[row, 0])), ('signal_shift', np.float32(t.scale_shift[row, 1])),
                 ('adapter_begin', np.uint32(int(record['seg_first'][adapter]) * pool)),
                 ('adapter_end', np.uint32((int(record['seg_last'][adapter]) + 1) * pool))]
        polya = t.polya_of(row)
        if polya is not None:
            tail = self.ctx.state_names.index('polya-tail') if 'polya-tail' in self.ctx.state_names else -1
            if tail >= 0 and record['seg_first'][tail] >= 0:
                attrs.append(('polya_end_debug', np.uint32((int(record['seg_last'][tail]) + 1) * pool)))
            attrs += [('polya_begin', np.uint32(polya['begin'])), ('polya_end', np.uint32(polya['end'])),
                      ('spikes', repr(polya['spikes']).encode())]
        read_id = t.read_id[row]
        if read_id not in self.event_dump_seen:       # (a second read of the same id keeps the first table)
            self.event_dump_seen.add(read_id)
            self.event_dump_list.append((read_id, table, attrs))

    def flush_event_dumps(self):
        """events/part-<worker>-<batch>.h5: basecalled_events/<batch>/<read id>, one compound
        dataset per read with its attributes (one file per batch: see flush_dumps)."""
        if not self.dump_events or self.event_dump_list is None:
            return
        from .fast5_write import H5Writer
        batch = self.formatted_dump_batchid
        top = os.path.join(self.outputdir, 'events')
        os.makedirs(top, exist_ok=True)
        with H5Writer(os.path.join(top, 'part-{}-{}.h5'.format(self.workerid, batch))) as h5:
            h5.require_group('basecalled_events/' + batch)
            for read_id, table, attrs in self.event_dump_list:
                h5.create_dataset('basecalled_events/{}/{}'.format(batch, read_id), table, attrs=attrs)
        self.event_dump_list = None


def albacore_frame(bcall):
    """Base-space view of a table that brings its own events (albacore's 14 columns), as load_events leaves
    it (signal_analyzer.py:311-326): start as stored, end = start + hstack(diff(start), [1]), pos =
    cumsum(move), p_model_state in the file's own float width."""
    ev = bcall.get('events')
    if ev is None:
        raise Exception('Unsupported event table found.')
    for name in ('start', 'mean', 'move'):
        if name not in ev:
            raise KeyError(name)
    start = np.asarray(ev['start'])
    signed = start.dtype.kind == 'i'
    st = start.astype(np.int64) if start.dtype.kind in 'iu' else start.astype(np.float64)
    if 'p_model_state' not in ev:
        pms = None
    else:
        pms = np.asarray(ev['p_model_state'])
    moves = np.asarray(ev['move'])
    end = np.append(st[1:], st[-1:] + 1) if len(st) else st
    return {'start': st, 'end': end, 'move': moves, 'pos': np.cumsum(moves), 'p_model_state': pms,
            'end_is_float': not signed, 'table': ev}


class SignalAnalysis:
    """One read's view of the batch (the reference's per-read object)."""

    def __init__(self, npread, analyzer):
        self.npread, self.config, self.analyzer = npread, analyzer.config, proxy(analyzer)

    def set_error(self, error):
        """`error`: a pack_unhandled_exception dict."""
        self.npread.set_error(error['status'], error['error_message'])

    def is_stopped(self):
        """True once a domain failure ended this read's processing."""
        return self.npread.is_stopped()

    def clear_cache(self):
        """Drop the raw samples and close the read's file."""
        self.npread.close()

    def process(self):
        """Single-read entry: the batch rules applied to this row alone."""
        self.analyzer.judge([self.npread.row], self.npread.table)

    def segments_of(self, record):
        """pxg_read_result -> {state name: (first, last)} (:354-362)."""
        names = self.analyzer.ctx.state_names
        first, last = record['seg_first'].tolist(), record['seg_last'].tolist()
        return {name: (first[i], last[i]) for i, name in enumerate(names) if first[i] >= 0}

    def load_events(self):
        if self.config['albacore_onthefly']:
            raise NotImplementedError('on-the-fly albacore basecalling is out of scope')
        bcall = self.npread.load_fast5_events()
        if self.npread.scaling_params is None:
            raise Exception('Signal scaling is not available yet.')
        if not self.config['filter_unsplit_reads'] and not (self.config['trim_adapter'] and self.config.get('trim_adapter_as_intended')):
            return bcall            # nothing downstream reads the table itself
        return self.event_frame(bcall)

    def event_frame(self, bcall):
        """Base-space columns of the Guppy event table (fast5_file.py:183-208,
        signal_analyzer.py:319-324).  The signal-space columns (mean, scaled_mean) never
        leave the GPU: SignalLoader.scan_unsplit_candidates."""
        if bcall.get('table') == 'albacore':             # the table's own events, unchanged (fast5_file.py:178-179)
            return albacore_frame(bcall)
        first, n_blocks, stride = self.npread.guppy_event_geometry(bcall=bcall)
        moves = np.asarray(bcall['move'], dtype=np.uint8)
        pos = np.cumsum(moves)
        if bcall.get('table') == 'guppy_events':         # Events table: the column is stored
            if bcall.get('p_model_state') is None:
                raise KeyError('p_model_state')
            pms = np.asarray(bcall['p_model_state'], dtype=np.float64)
        else:                                            # Move table: from the quality string
            lead = {5: 2, 1: 0}.get(len(bcall['sequence']) - (int(pos[-1]) if len(pos) else 0) + 1)
            if lead is None:
                raise Exception('Move table is encoded with an unknown kmer-size.')
            phred = np.frombuffer(bcall['qstring'].encode(), 'B') - 33
            pms = (1 - 10 ** -(phred / 10))[pos - 1 + lead]
        start = first + stride * np.arange(n_blocks, dtype=np.int64)
        end = start
        if n_blocks:                                     # (the next event's start; one sample for the last)
            end = np.empty(n_blocks, dtype=np.int64)
            end[:-1], end[-1] = start[1:], start[-1] + 1
        return {'start': start, 'end': end, 'move': moves, 'pos': pos, 'p_model_state': pms}

    def trim_adapter(self, events, segments, elspan):
        """signal_analyzer.py:328-344.  The reference returns as soon as a sequence IS present (:329-331) -- which is
        always the case after load_events: `--trim-adapter` is a no-op in this revision of the reference (SURVEY section
        0), and stays one here by default.  `config['trim_adapter_as_intended'] = True` (not a reference option) runs
        what the rest of that function was written to do, with the test the other way round: the basecalled length of
        the adapter = moves of the events that start at or before the adapter's last sample + the k-mer's leading
        half, recorded as the read's trimming length; longer than the sequence: 'basecall_table_incomplete'."""
        if not self.config.get('trim_adapter_as_intended'):
            return
        sequence = self.npread.sequence
        if sequence is None or 'adapter' not in segments:
            return
        adapter_end = segments['adapter'][1] * elspan
        kmer_lead_size = self.analyzer.kmersize // 2
        in_adapter = np.asarray(events['start']) <= adapter_end          # (:334, row by row: an albacore table need not be sorted)
        if not in_adapter.any():
            return
        adapter_basecall_length = int(np.asarray(events['move'], dtype=np.int64)[in_adapter].sum()) + kmer_lead_size
        if adapter_basecall_length > len(sequence[0]):
            raise SignalAnalysisError('basecall_table_incomplete')
        if adapter_basecall_length > 0:
            self.npread.set_adapter_trimming_length(adapter_basecall_length)

    def detect_unsplit_read(self, events, segments, elspan):
        """Decision rule of :420-443 over the in-read adapter candidates the GPU window
        scan found for this read: count confidently called bases in the stretches between
        the candidates and compare the later sub-reads with the first."""
        t, row = self.npread.table, self.npread.row
        if 'adapter' not in segments:
            return False            # must be an adapter-only read
        if t.unsplit_count[row] < 0:
            raise Exception('chimera window scan failed for this read (code {})'.format(
                int(t.unsplit_count[row])))
        if events.get('end_is_float'):
            # :385 range(payload_start, events.iloc[-1]['end'], window_step): `end' = start + duration is a
            # float64 column whenever `start' is not a signed integer column (uint64 + int64 promotes, and
            # albacore writes uint64), and range() refuses it -- the reference fails this read right here
            raise TypeError("'numpy.float64' object cannot be interpreted as an integer")
        if events.get('table') is not None and row not in getattr(t, 'own_table_scanned', ()):
            # an albacore table the window scan did not take (a float64 `mean' column, starts that are not
            # ascending): refused for this read alone, never silently passed
            raise Exception('Unsupported event table found.')
        candidates = t.unsplit[row]
        if not candidates:
            return False
        limits = self.config['unsplit_read_detection']
        payload_start = (segments['adapter'][1] + 1) * elspan
        cuts = np.array([[0, payload_start]] + union_intervals(candidates) + [[np.inf, np.inf]])
        # sub-read k = events whose start lies in [cuts[k, 1], cuts[k + 1, 0]], both inclusive
        start, pos, pms = events['start'], events['pos'], events['p_model_state']
        if pms is None:
            raise KeyError('p_model_state')
        lo = np.searchsorted(start, cuts[:-1, 1], side='left')
        hi = np.searchsorted(start, cuts[1:, 0], side='right')
        # a base (one value of `pos`) counts when the best p_model_state among its events
        # INSIDE the sub-read clears the limit (a cut may split a base's events)
        hq = []
        for a, b in zip(lo.tolist(), hi.tolist()):
            if b <= a:
                hq.append(0)
                continue
            p = pos[a:b]
            first_of_base = np.empty(b - a, dtype=bool)          # (np.r_[True, p[1:] != p[:-1]] without the index trick)
            first_of_base[0] = True
            np.not_equal(p[1:], p[:-1], out=first_of_base[1:])
            heads = np.nonzero(first_of_base)[0]
            # (a float32 column is compared in float32, as pandas / NumPy 1.x compare it with a Python float)
            limit = pms.dtype.type(limits['basecount_quality_limit'])
            hq.append(int((np.maximum.reduceat(pms[a:b], heads) > limit).sum()))
        later = sum(hq[1:])
        return bool(later > limits['subread_basecount_limit'] or
                    (later + 1) / (hq[0] + 1) > limits['subread_baseratio_limit'])

    def detect_segments(self, signal, elspan):
        """Single-read debug path (:346-364) through the GPU hook."""
        scan_limit = self.config['segmentation']['segmentation_scan_limit'] // elspan
        first, last, _, _ = self.analyzer.ctx.viterbi([signal[:scan_limit]])
        names = self.analyzer.ctx.state_names
        return {names[i]: (int(first[0][i]), int(last[0][i]))
                for i in range(len(names)) if first[0][i] >= 0}

    def push_barcode_signal(self, signal, segments):
        self.analyzer.demuxer.push(self.npread, signal)
