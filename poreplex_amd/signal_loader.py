"""Batch read table and loader (reference module: poreplex/signal_loader.py).

The reference keeps one ``NanoporeRead`` object per read and walks them in
Python between two Keras ``predict`` calls (signal_loader.py:77-109,112-198).
A GPU worker batch is thousands of reads, so here a batch is ONE columnar
``ReadTable`` (NumPy columns for everything numeric, the GPU's
``pxg_read_result`` array attached as it comes back) and every rule that does
not need a per-read file access is a column operation.  ``NanoporeRead`` is
kept as the reference's operator surface -- same constructor-less attribute
and method names, same ``report()`` dict (keys and order of
signal_loader.py:165-198) -- but it is only a (table, row) handle.

``SignalLoader.fit_scalers`` is the single GPU pass of the batch: scaler
network + QC, pooling, segmentation, barcode classifier, optional poly(A) and
the chimera window scan all run behind one upload.
"""
import gc
import os
import threading
from operator import itemgetter

import numpy as np

from . import native
from .fast5_file import ReadBundle, open_read

__all__ = ['SignalLoader', 'NanoporeRead', 'ReadTable', 'SignalAnalysisError']

LABELS = ('pass', 'fail', 'artifact')
_OKAY = native.STATUS_CODE['okay']
_NO_LABEL = -1


class SignalAnalysisError(Exception):
    pass


def _grown(col, n):
    """`col` with room for at least n rows (amortised doubling, zero filled)."""
    if n <= len(col):
        return col
    out = np.zeros((max(n, 2 * len(col), 64),) + col.shape[1:], dtype=col.dtype)
    out[:len(col)] = col
    return out


class ReadTable:
    """Struct of arrays for the reads of one worker batch that opened."""

    NUMERIC = (('status', np.int8), ('stopped', np.bool_), ('label', np.int8),
               ('barcode', np.int8), ('barcode_guess', np.int8), ('barcode_phred', np.int16),
               ('has_barcode', np.bool_), ('has_scaling', np.bool_), ('has_summary', np.bool_),
               ('start_time', np.int64), ('duration', np.int64), ('sampling_rate', np.float64),
               ('num_events', np.int64), ('sequence_length', np.int64),
               ('mean_qscore', np.float64), ('n_raw', np.int64), ('pending', np.bool_),
               ('seq_lazy', np.bool_),
               ('unsplit_count', np.int32),
               # poly(A) tail as the GPU record has it; the dict of polya.py:116-121 is built on demand
               ('polya_lazy', np.bool_), ('polya_begin', np.int64), ('polya_end', np.int64),
               ('polya_dwell_time', np.float64), ('polya_spike_count', np.int32))

    def __init__(self, capacity=0):
        """`capacity`: rows to make room for up front (a worker call knows how many reads it brings: the columns are
        then allocated once instead of grown from nothing)."""
        self.n = 0
        for name, dtype in self.NUMERIC:
            setattr(self, name, np.zeros(capacity, dtype=dtype))
        self.scale_shift = np.zeros((capacity, 2), dtype=np.float32)
        self.calib = np.zeros(capacity, dtype=native.CALIB_DTYPE)
        # per-row Python objects
        self.filename, self.read_id, self.source = [], [], []
        self.channel, self.run_id, self.sample_id = [], [], []
        self.raw, self.sequence, self.error_message, self.polya = [], [], [], []
        self.unsplit = []
        # rows that came out of a read bundle: the bundle and the read's index in it
        self.bundle, self.bundle_index = None, np.zeros(capacity, dtype=np.int64)
        # attached by the GPU pass
        self.gpu_row = np.zeros(capacity, dtype=np.int64)      # row -> index in `records`, -1 if not run
        self.records = None
        self.adapter_dump = None      # --dump-adapter-signals: (values, offsets) by GPU row
        self.event_frame = self.event_dump = None   # --dump-basecalls: frames by GPU row, columns by block stride
        self.spikes = None            # poly(A) spike rows of the GPU pass [total, 4] ...
        self.spike_offsets = None     # ... rows of GPU record g: spikes[spike_offsets[g]:spike_offsets[g + 1]]
        self._opened = False          # some row holds its own samples / an open file

    def append(self, filename, read_id, source):
        i = self.n
        self._opened = True
        self.n = i + 1
        for name, _ in self.NUMERIC:
            setattr(self, name, _grown(getattr(self, name), self.n))
        self.scale_shift = _grown(self.scale_shift, self.n)
        self.calib = _grown(self.calib, self.n)
        self.gpu_row = _grown(self.gpu_row, self.n)
        self.bundle_index = _grown(self.bundle_index, self.n)
        self.bundle_index[i] = getattr(source, 'i', -1) if getattr(source, 'bundle', None) is not None else -1
        if self.bundle_index[i] >= 0:
            self.bundle = source.bundle
        self.status[i], self.label[i], self.gpu_row[i] = _OKAY, _NO_LABEL, -1
        self.start_time[i], self.duration[i] = source.start_time, source.duration
        self.sampling_rate[i] = source.sampling_rate
        self.calib[i] = (source.range, source.digitization, source.offset, source.sampling_rate)
        self.filename.append(filename)
        self.read_id.append(read_id)
        self.source.append(source)
        self.channel.append(source.channel_number)
        self.run_id.append(source.run_id)
        self.sample_id.append(source.sample_id)
        for col in (self.raw, self.sequence, self.error_message, self.polya, self.unsplit):
            col.append(None)
        return i

    def extend_from_bundle(self, bundle, idx):
        """Append the bundle's reads `idx` (int array) in one go: every column is a slice of
        the bundle's columns, the raw samples are views into its arena.  Returns the rows."""
        idx = np.asarray(idx, dtype=np.int64)
        k, d = len(idx), bundle.d
        lo, hi = self.n, self.n + k
        self.n = hi
        if hi > len(self.status):               # (a table made with room for its batch skips this)
            for name, _ in self.NUMERIC:
                setattr(self, name, _grown(getattr(self, name), hi))
            for name in ('scale_shift', 'calib', 'gpu_row', 'bundle_index'):
                setattr(self, name, _grown(getattr(self, name), hi))
        rows = np.arange(lo, hi)
        run = bool(k and idx[-1] - idx[0] == k - 1 and (k == 1 or (np.diff(idx) == 1).all()))
        # (a run of the bundle's reads -- the usual batch -- is slices on both sides, not 10 000-element index arrays)
        at, src = (slice(lo, hi), slice(int(idx[0]), int(idx[0]) + k)) if run else (rows, idx)
        self.status[at], self.label[at], self.gpu_row[at] = _OKAY, _NO_LABEL, -1
        self.bundle, self.bundle_index[at] = bundle, idx
        self.start_time[at], self.duration[at] = d['start_time'][src], d['duration'][src]
        self.calib[at] = d['calib'][src]
        self.sampling_rate[at] = d['calib']['sampling_rate'][src]
        o = d['offsets']
        self.n_raw[at] = (o[int(idx[0]) + 1:int(idx[0]) + k + 1] - o[int(idx[0]):int(idx[0]) + k]) if run else o[idx + 1] - o[idx]
        if run:
            self.filename += bundle.filenames[int(idx[0]):int(idx[-1]) + 1]
            self.read_id += bundle.read_ids[int(idx[0]):int(idx[-1]) + 1]
        else:
            self.filename += [bundle.filenames[i] for i in idx.tolist()]
            self.read_id += [bundle.read_ids[i] for i in idx.tolist()]
        texts = bundle.text_lists(idx) if hasattr(bundle, 'text_lists') else None
        if texts is not None:             # (the lists kept with the open files: no new str per read and batch)
            self.channel += texts[0]
            self.run_id += texts[1]
            self.sample_id += texts[2]
        else:
            self.channel += d['channel_number'][idx].tolist()
            self.run_id += d['run_id'][idx].tolist()
            self.sample_id += d['sample_id'][idx].tolist()
        self.pending[at] = True              # samples wait in the bundle arena (raw stays None)
        for col in (self.raw, self.source, self.sequence, self.error_message, self.polya, self.unsplit):
            col.extend([None] * k)
        return rows

    def polya_of(self, i):
        """set_polya_tail's dict for row i (None if no tail was called)."""
        if self.polya[i] is None and self.polya_lazy[i]:
            ns = int(self.polya_spike_count[i])
            spikes = []
            if ns and self.spikes is not None:
                at = int(self.spike_offsets[self.gpu_row[i]])
                spikes = [tuple(r) for r in self.spikes[at:at + ns].astype(float).tolist()]
            self.polya[i] = {'begin': int(self.polya_begin[i]), 'end': int(self.polya_end[i]),
                             'dwell_time': float(self.polya_dwell_time[i]), 'spikes': spikes}
            self.polya_lazy[i] = False
        return self.polya[i]

    def samples_of(self, i):
        """int16 samples of a read that has not been packed yet."""
        if self.raw[i] is not None:
            return self.raw[i]
        return self.bundle.samples(int(self.bundle_index[i]))

    def sequence_of(self, i):
        """(sequence, quality string, adapter trim length) or None; bundle rows settled by the
        bulk rules keep theirs in the bundle until somebody asks (FASTQ output, result dicts)."""
        if self.sequence[i] is None and self.seq_lazy[i]:
            seq, qual = self.bundle.sequence_of(int(self.bundle_index[i]))
            self.sequence[i] = (seq, qual, 0)
        return self.sequence[i]

    def source_of(self, i):
        """The read's file object; bundle rows get theirs on first use."""
        if self.source[i] is None and self.bundle_index[i] >= 0:
            from .fast5_file import BundleReader
            self.source[i] = BundleReader(self.bundle, int(self.bundle_index[i]))
        return self.source[i]

    # -- column updates ------------------------------------------------------
    def halt(self, rows, status, label=None):
        """Domain failure of `rows`: status string, processing stops, optional label."""
        self.status[rows] = native.STATUS_CODE[status]
        self.stopped[rows] = True
        if label is not None:
            self.label[rows] = LABELS.index(label)

    def live_rows(self, rows=None):
        rows = np.arange(self.n) if rows is None else np.asarray(rows, dtype=np.int64)
        return rows[~self.stopped[rows]]

    def record_of(self, row):
        g = self.gpu_row[row]
        return None if g < 0 or self.records is None else self.records[g]

    def release(self, rows):
        self.pending[np.asarray(rows, dtype=np.int64)] = False
        if not self._opened:          # nothing but bundle rows: no samples held, no file open
            return
        for i in rows:
            self.raw[i] = None
            src = self.source[i]
            if src is not None:
                src.close()

    # -- result dicts --------------------------------------------------------
    def report(self, rows):
        """Result dicts of `rows`, keys and key order of signal_loader.py:165-198.  Built by
        the host extension (csrc/pxg_pyreport.c) straight from the columns when it is there;
        the loop below is the same thing in Python."""
        idx = np.ascontiguousarray(rows, dtype=np.int64)
        fast = native.load_pyhost()
        if fast is not None:
            # ten thousand dicts + tuples in one C call: none of them can be part of a cycle, and the
            # collector would walk the young ones some thirty times on the way (and, now and then,
            # everything the process holds)
            was_on = gc.isenabled()
            gc.disable()
            try:
                return fast.report(self._report_columns(), idx)
            finally:
                if was_on:
                    gc.enable()
        rows = idx.tolist()
        status = [native.STATUS_NAMES[c] for c in self.status[idx].tolist()]
        # Python's round (correctly rounded decimal), not np.round: part of the output contract
        start = [round(a / b, 3) for a, b in zip(self.start_time[idx].tolist(),
                                                 self.sampling_rate[idx].tolist())]
        duration = self.duration[idx].tolist()
        n_events = self.num_events[idx].tolist()
        seq_len = self.sequence_length[idx].tolist()
        # the reference's default for a read without a basecall summary is the int 0
        qscore = [q if got else 0 for q, got in zip(self.mean_qscore[idx].tolist(),
                                                    self.has_summary[idx].tolist())]
        label = self.label[idx].tolist()
        called = self.has_barcode[idx].tolist()
        barcode = self.barcode[idx].tolist()
        guess = self.barcode_guess[idx].tolist()
        phred = self.barcode_phred[idx].tolist()
        out = []
        for k, i in enumerate(rows):
            rep = {'filename': self.filename[i], 'read_id': self.read_id[i], 'status': status[k],
                   'channel': self.channel[i], 'start_time': start[k], 'run_id': self.run_id[i],
                   'sample_id': self.sample_id[i], 'duration': duration[k],
                   'num_events': n_events[k], 'sequence_length': seq_len[k],
                   'mean_qscore': qscore[k]}
            if self.sequence_of(i) is not None:
                rep['sequence'] = self.sequence[i]
            if self.error_message[i]:
                rep['error_message'] = self.error_message[i]
            if label[k] != _NO_LABEL:
                rep['label'] = LABELS[label[k]]
            if called[k]:
                rep['barcode'], rep['barcode_guess'] = barcode[k], guess[k]
                rep['barcode_score'] = phred[k]
            if self.polya_of(i) is not None:
                rep['polya'] = self.polya[i]
            out.append(rep)
        return out


def _report_columns(self):
    """The columns pxg_pyreport.c reads (contiguous NumPy arrays / the per-row lists)."""
    c = {name: np.ascontiguousarray(getattr(self, name)) for name in (
        'status', 'start_time', 'sampling_rate', 'duration', 'num_events', 'sequence_length',
        'mean_qscore', 'has_summary', 'label', 'has_barcode', 'barcode', 'barcode_guess',
        'barcode_phred', 'seq_lazy', 'bundle_index', 'polya_lazy', 'polya_begin', 'polya_end',
        'polya_dwell_time', 'polya_spike_count', 'gpu_row')}
    for name in ('filename', 'read_id', 'channel', 'run_id', 'sample_id', 'sequence',
                 'error_message', 'polya'):
        c[name] = getattr(self, name)
    c['status_names'], c['label_names'] = native.STATUS_NAMES, LABELS
    if self.bundle is not None:
        d = self.bundle.d
        c['seq_arena'], c['qual_arena'] = d['seq_arena'], d['qual_arena']
        c['seq_offsets'] = np.ascontiguousarray(d['seq_offsets'], dtype=np.int64)
    if self.spikes is not None:
        c['spikes'] = np.ascontiguousarray(self.spikes, dtype=np.float32)
        c['spike_offsets'] = np.ascontiguousarray(self.spike_offsets, dtype=np.int64)
    return c


ReadTable._report_columns = _report_columns


def summary_columns(table, rows, barcoding, polya):
    """The values SequencingSummaryWriter prints for `rows`, as per-field lists of the SAME
    Python objects report() would put into the dicts (so str() of them is identical), without
    building the dicts.  Only labelled rows are written by the writer: the caller filters."""
    idx = np.asarray(rows, dtype=np.int64)
    rows = idx.tolist()

    def pick(column):                      # column[i] for i in rows, at C speed
        if not rows:
            return []
        got = itemgetter(*rows)(column)
        return list(got) if len(rows) > 1 else [got]
    cols = {
        'filename': pick(table.filename), 'read_id': pick(table.read_id),
        'run_id': pick(table.run_id), 'channel': pick(table.channel),
        'start_time': [round(a / b, 3) for a, b in zip(table.start_time[idx].tolist(),
                                                       table.sampling_rate[idx].tolist())],
        'duration': table.duration[idx].tolist(),
        'num_events': table.num_events[idx].tolist(),
        'sequence_length': table.sequence_length[idx].tolist(),
        'mean_qscore': [q if got else 0 for q, got in zip(table.mean_qscore[idx].tolist(),
                                                          table.has_summary[idx].tolist())],
        'sample_id': pick(table.sample_id),
        'status': [native.STATUS_NAMES[c] for c in table.status[idx].tolist()],
        'label': [LABELS[c] for c in table.label[idx].tolist()],
    }
    if barcoding:
        called = table.has_barcode[idx]
        cols['barcode'] = [b if c else None for b, c in zip(table.barcode[idx].tolist(), called.tolist())]
        cols['barcode_score'] = np.where(called, table.barcode_phred[idx], 0).tolist()
    if polya:        # dwell time in seconds, None where no tail was called
        lazy = table.polya_lazy[idx].tolist()
        cols['polya_dwell_time'] = [d if z else (p['dwell_time'] if p is not None else None)
                                    for d, z, p in zip(table.polya_dwell_time[idx].tolist(), lazy,
                                                       pick(table.polya))]
    return cols


class NanoporeRead:
    """Row handle with the reference's NanoporeRead surface."""

    __slots__ = ('table', 'row')

    def __init__(self, table, row):
        self.table, self.row = table, row

    # identity / metadata
    filename = property(lambda self: self.table.filename[self.row])
    read_id = property(lambda self: self.table.read_id[self.row])
    fast5 = property(lambda self: self.table.source_of(self.row))
    sampling_rate = property(lambda self: float(self.table.sampling_rate[self.row]))
    status = property(lambda self: native.STATUS_NAMES[self.table.status[self.row]])
    stopped = property(lambda self: bool(self.table.stopped[self.row]))
    error_message = property(lambda self: self.table.error_message[self.row])
    sequence = property(lambda self: self.table.sequence_of(self.row))
    polya = property(lambda self: self.table.polya_of(self.row))
    num_events = property(lambda self: int(self.table.num_events[self.row]))
    sequence_length = property(lambda self: int(self.table.sequence_length[self.row]))
    mean_qscore = property(lambda self: float(self.table.mean_qscore[self.row])
                           if self.table.has_summary[self.row] else 0)
    native = property(lambda self: self.table.record_of(self.row))

    @property
    def label(self):
        code = self.table.label[self.row]
        return None if code == _NO_LABEL else LABELS[code]

    @property
    def scaling_params(self):
        t = self.table
        return t.scale_shift[self.row].copy() if t.has_scaling[self.row] else None

    @property
    def barcode(self):
        t = self.table
        return int(t.barcode[self.row]) if t.has_barcode[self.row] else None

    @property
    def native_spikes(self):
        t, g = self.table, self.table.gpu_row[self.row]
        if t.spikes is None or g < 0:
            return None
        return t.spikes[int(t.spike_offsets[g]):int(t.spike_offsets[g + 1])]

    # setters of the reference surface
    def set_status(self, newstatus, stop=False):
        t = self.table
        t.status[self.row] = native.STATUS_CODE[newstatus]
        t.stopped[self.row] |= bool(stop)

    def set_error(self, status, error_message):
        self.table.status[self.row] = native.STATUS_CODE[status]
        self.table.error_message[self.row] = error_message

    def set_scaling_params(self, params):
        self.table.scale_shift[self.row] = params
        self.table.has_scaling[self.row] = True

    def set_label(self, newlabel):
        self.table.label[self.row] = LABELS.index(newlabel)

    def set_barcode(self, newbarcode, guess, quality):
        t, i = self.table, self.row
        t.has_barcode[i] = newbarcode is not None
        t.barcode[i] = -1 if newbarcode is None else newbarcode
        t.barcode_guess[i], t.barcode_phred[i] = guess, quality

    def set_adapter_trimming_length(self, newlength):
        seq = self.table.sequence_of(self.row)
        if seq is None:
            raise Exception('Sequence is not set.')
        self.table.sequence[self.row] = (seq[0], seq[1], newlength)

    def set_polya_tail(self, polya_info):
        self.table.polya[self.row] = polya_info
        self.table.polya_lazy[self.row] = False

    def is_stopped(self):
        return self.stopped

    def close(self):
        self.table.release([self.row])

    def report(self):
        return self.table.report([self.row])[0]

    # basecall access (the FAST5 is read per read; nothing to batch)
    def load_fast5_events(self):
        """Basecall summary into the table (signal_loader.py:266-279).  The reference
        builds the whole event table inside get_basecall, so a basecall group whose table
        is missing, of an unknown kind or of the wrong size fails BEFORE any summary field
        is stored (fast5_file.py:166-181,210-223): validate first, commit second.  The
        table's signal columns are only materialised by the stage that consumes them."""
        t, i = self.table, self.row
        if t.source_of(i) is None:
            raise Exception('Fast5 must be open for getting events.')
        source = t.source[i]
        # (a bundle row hands its Move column over as an array: nothing below needs the list)
        bcall = source.get_basecall(move_as_array=True) if t.bundle_index[i] >= 0 and t.bundle is not None \
            else source.get_basecall()
        if bcall is None:
            raise SignalAnalysisError('not_basecalled')
        kind = bcall.get('table', 'move' if bcall.get('move') is not None else None)
        if kind is None:
            raise Exception("Neither `Events' or `Move' table found in the basecall.")
        if kind == 'unsupported':
            raise Exception('Unsupported event table found.')
        if kind != 'albacore':          # Guppy frames are re-cut from the raw signal
            self.guppy_event_geometry(bcall=bcall)
        t.sequence_length[i], t.mean_qscore[i] = bcall['sequence_length'], bcall['mean_qscore']
        t.num_events[i], t.has_summary[i] = bcall['num_events'], True
        t.sequence[i] = (bcall['sequence'], bcall['qstring'], 0)
        return bcall

    def guppy_event_geometry(self, n_raw=None, bcall=None):
        """(first_sample, n_blocks, block_stride) of the Guppy event frame under the size
        rule of convert_events_guppy (fast5_file.py:210-223): the raw slice, padded to
        whole blocks, must hold exactly one block per move."""
        t, i = self.table, self.row
        if bcall is None:
            bcall = t.source_of(i).get_basecall()
        if bcall is None:
            raise SignalAnalysisError('not_basecalled')
        if bcall.get('move') is None:
            raise Exception("Neither `Events' or `Move' table found in the basecall.")
        if bcall.get('table') == 'albacore':        # (its events are its own: SignalLoader.unsplit_event_tables)
            raise Exception('an albacore Events table has no Guppy block frame')
        n_raw = int(t.n_raw[i]) if n_raw is None else int(n_raw)
        first, stride = int(bcall['first_sample_template']), int(bcall['block_stride'])
        n_blocks = len(bcall['move'])
        covered = max(min(first + stride * n_blocks, n_raw) - first, 0)
        if -(-covered // stride) != n_blocks:
            raise Exception('Numbers of events and raw data strides does not match.')
        return first, n_blocks, stride


class CallArenas:
    """int16 sample arenas for the per-call bundles of worker calls that read FAST5 files (SignalLoader.fast5_run_bundle):
    a 128-read call decodes ~8 MB of samples, and memory that comes fresh from the allocator costs a page fault per
    4 KB on the way in (0.72 -> 0.55 ms for the samples of such a call on the development host).  A call takes one,
    gives it back when its records are down; calls on other threads find it there.  At most `keep` arenas wait.
    With `ctx` (PXG_PIN_CALL_ARENAS=1, an experiment that has not been on a GPU yet: off by default) the arenas are
    mappings of their own, page-locked once, so a call's samples cross PCIe as one DMA transfer instead of through the
    context's 8 MB chunks; release() takes the page locks off."""

    def __init__(self, keep=64, ctx=None):
        self.free, self.keep, self.lock = [], keep, threading.Lock()
        self.ctx, self.locked = ctx, []

    def take(self, n_samples):
        with self.lock:
            for k in range(len(self.free) - 1, -1, -1):          # the one given back last is the warmest
                if len(self.free[k]) >= n_samples:
                    return self.free.pop(k)
            if self.free and self.ctx is None:
                self.free.pop(0)                                  # (too small for today's calls: make room for one that fits)
        size = max(int(n_samples) + int(n_samples) // 4, 1 << 20)
        if self.ctx is None:
            return np.empty(size, dtype=np.int16)
        arena = native.page_exclusive(size, np.int16)
        self.ctx.pin(arena)
        with self.lock:
            self.locked.append(arena)
        return arena

    def give(self, arena):
        with self.lock:
            if len(self.free) < self.keep or self.ctx is not None:      # (a page-locked arena is never just dropped)
                self.free.append(arena)

    def release(self):
        with self.lock:
            locked, self.locked, self.free = self.locked, [], []
        for arena in locked:
            try:
                self.ctx.unpin(arena)
            except Exception:             # noqa: BLE001  (the context is on its way out)
                pass


class CallBundle:
    """The reads of one worker call over FAST5 files (SignalLoader.fast5_run_plan): `layout` -- where their samples and
    text go (fast5_file.Fast5Batch.plan) --, `plain` -- the columns the short path reads, the call's reads at rows
    [first, first + n) --, `arena` -- the pooled sample arena --, and bundle(): the ReadBundle over exactly these reads,
    made when somebody needs one (the batch table of a call that declines, reads with chimera candidates)."""

    def __init__(self, runs, layout, plain, first, arena):
        self.runs, self.layout, self.plain, self.first, self.arena = runs, layout, plain, first, arena
        self.made = None

    def bundle(self):
        if self.made is None:
            from .fast5_file import Fast5Batch
            self.made = Fast5Batch.from_runs(self.runs).bundle(self.layout)
        return self.made


class _Decoding:
    def __init__(self, loader):
        self.loader = loader

    def __enter__(self):
        with self.loader._stage_lock:
            self.loader._decoding += 1
            return None if self.loader._decoding == 1 else 1

    def __exit__(self, *exc):
        with self.loader._stage_lock:
            self.loader._decoding -= 1
        return False


class SignalLoader:
    """Opens reads into a ReadTable and runs the GPU pass over it.  `self.table` is the
    batch the reference-style calls (prepare_loading / fit_scalers) work on; the session
    driver keeps several tables in flight and passes them explicitly."""

    def __init__(self, config, fast5prefix, ctx, read_bundle=None):
        self.config, self.fast5prefix, self.ctx = config, fast5prefix, ctx
        self.bundle = ReadBundle(read_bundle) if read_bundle else None
        c = ctx.cfg
        self.scaler_cfg = {        # scaler-r3 attrs (signal_loader.py:55-58)
            'dtype': 'float32', 'stride': int(c.stride), 'length': int(c.scaler_length),
            'min_length': int(c.scaler_min_length),
            'qc_scale': (float(c.scaler_qc_scale[0]), float(c.scaler_qc_scale[1])),
            'qc_shift': (float(c.scaler_qc_shift[0]), float(c.scaler_qc_shift[1])),
        }
        self.stage_mask = native.STAGE_ALL_DEMUX
        self.scan_unsplit = False      # --filter-chimera: also run the a19 window scan
        self.dump_adapter = False      # --dump-adapter-signals: the adapter stretch comes back with the records
        self.dump_events = False       # --dump-basecalls: so do mean / stdv / scaled mean of every Guppy block
        self.table = ReadTable()
        # several worker calls may be in flight on one context (threads: fit_scalers): one call
        # owns the spare input slot from stage to swap, one owns the resident batch from swap
        # to the download of its records
        self._stage_lock, self._run_lock = threading.Lock(), threading.Lock()
        self._pinned = []
        self._decoding = 0             # worker calls inside fast5_run_bundle right now
        self.call_arenas = CallArenas(ctx=ctx if os.environ.get('PXG_PIN_CALL_ARENAS') and hasattr(ctx, 'pin') else None)

    def clear(self):
        t = self.table
        if t.n or t.records is not None or t._opened:      # (an untouched table is as good as a new one)
            self.table = ReadTable()

    def exists(self, filename):
        if self.bundle is not None and self.bundle.has_file(filename):
            return True
        return os.path.exists(os.path.join(self.fast5prefix, filename))

    def prepare_loading(self, filename, read_id, table=None):
        """Open one read into the batch table.  An unreadable file raises: the reference
        marks it 'irregular_fast5' (signal_loader.py:200-207) and then trips over the
        missing reader, so its caller reports 'unknown_error' -- same outcome here."""
        source = open_read(os.path.join(self.fast5prefix, filename), filename, read_id,
                           self.bundle)
        t = self.table if table is None else table
        row = t.append(filename, read_id, source)
        raw = np.ascontiguousarray(source.get_raw_int16(), dtype=np.int16)
        t.n_raw[row] = len(raw)
        cfg = self.scaler_cfg      # length gate of load_padded_signal_head (:212-222)
        usable = min(cfg['length'], int(t.duration[row]), len(raw))
        if usable - usable % cfg['stride'] < cfg['min_length']:
            t.halt(row, 'scaler_signal_too_short')
        else:
            t.raw[row], t.pending[row] = raw, True
        return NanoporeRead(t, row)

    def prepare_many(self, reads, table, reserve=None, prebuilt=None):
        """Bulk form of prepare_loading: reads that live in the read bundle are one column
        append; reads that live in FAST5 files become the same columns through the native
        reader (fast5_file.Fast5Batch: metadata per file, signals and basecall text decoded on
        host threads -- into `reserve(n_samples)`, the session's staging arena, when given).
        Returns an int array with one entry per input read: its row, or -1 when the read needs
        the per-read path (file gone or unreadable, read not in its file, bundle marked corrupt:
        that path raises / reports exactly what the reference does)."""
        b = self.bundle
        where = np.full(len(reads), -1, dtype=np.int64)
        if not len(reads):
            return where
        if b is None or not b.has_file(reads[0][0]):
            return self.prepare_fast5(reads, where, table, reserve, prebuilt)
        index, broken = b.index, b.broken
        # the usual worker batch is a run of consecutive bundle reads: one list comparison
        # instead of a dictionary lookup per read
        first = index.get(reads[0], -1)
        if first >= 0 and b.keys[first:first + len(reads)] == reads:
            where[:] = np.arange(first, first + len(reads))
        else:
            where[:] = [index.get(key, -1) for key in reads]
        if broken:
            where[[key[0] in broken for key in reads]] = -1
        found = where >= 0
        rows = table.extend_from_bundle(b, where[found])
        cfg = self.scaler_cfg      # length gate of load_padded_signal_head (:212-222)
        usable = np.minimum(np.minimum(cfg['length'], table.duration[rows]), table.n_raw[rows])
        short = rows[usable - usable % cfg['stride'] < cfg['min_length']]
        table.halt(short, 'scaler_signal_too_short')
        table.pending[short] = False
        where[found] = rows
        return where

    def prefetch_files(self, reads):
        """Open the FAST5 files of a coming batch (handle, read ids, metadata columns: all cached by
        fast5_file.open_fast5) -- the session's loader thread runs this beside the batch it is
        decoding.  Errors are left to prepare_many, which reports them per read."""
        from .fast5_file import file_run_columns, open_fast5
        if not reads or (self.bundle is not None and self.bundle.has_file(reads[0][0])):
            return
        files = dict.fromkeys(key[0] for key in reads)
        if len(files) == len(reads) and len(files) >= self.SINGLE_READ_BATCH_MIN:
            return            # one file per read: the batch opens them in one native call (prepare_single_read_files)
        for filename in files:
            try:
                f = open_fast5(os.path.join(self.fast5prefix, filename))
                f.read_ids, f.info
                if f.multi:
                    f.read_ids_array, f.keys_for(filename)
                    file_run_columns(f, filename)            # (what the batch's bundle slices its metadata columns from)
            except Exception:             # noqa: BLE001
                pass

    def prepare_fast5(self, reads, where, table, reserve=None, prebuilt=None):
        """The FAST5 half of prepare_many.  Only when the table holds no bundle rows yet (a
        table has one column source).  `prebuilt`: the decoded CallBundle of exactly these reads (fast5_run_plan)."""
        from .fast5_file import Fast5Batch, Fast5Error, open_fast5
        if table.n or table.bundle is not None:
            return where
        if prebuilt is not None and prebuilt.layout['n'] == len(reads):
            prebuilt = prebuilt.bundle()
            return self.enter_fast5_bundle(prebuilt, prebuilt.filenames, np.arange(len(reads)), where, table)
        runs = self.fast5_runs(reads)
        if runs is not None:
            bundle = Fast5Batch.from_runs(runs).as_bundle(reserve)
            return self.enter_fast5_bundle(bundle, bundle.filenames, np.arange(len(reads)), where, table)
        done = self.prepare_single_read_files(reads, where, table, reserve)
        if done is not None:
            return done
        # per FILE, not per read: the positions of its reads in the request, their indices in
        # the file by one dictionary pass, the readable ones by one mask over the info column
        by_file = {}
        for pos, key in enumerate(reads):
            by_file.setdefault(key[0], []).append(pos)
        files, index, at = [], [], []
        for filename, positions in by_file.items():
            try:
                f = open_fast5(os.path.join(self.fast5prefix, filename))
            except (OSError, Fast5Error):
                continue                         # vanished or unreadable: the per-read path says how
            if f.multi:
                f.read_ids
                lookup = f._index
                i = np.array([lookup.get(reads[pos][1], -1) for pos in positions], dtype=np.int64)
            else:
                first = f.read_ids[0] if f.n else None
                i = np.array([0 if reads[pos][1] == first else -1 for pos in positions], dtype=np.int64)
            ok = i >= 0
            ok[ok] = f.info['status'][i[ok]] == 0
            if ok.any():
                files.append((f, int(ok.sum())))
                index.append(i[ok])
                at.append(np.asarray(positions, dtype=np.int64)[ok])
        if not files:
            return where
        # (request order is kept within a file; files come in the order of their first read)
        at, index = np.concatenate(at), np.concatenate(index)
        names = [reads[pos][0] for pos in at.tolist()]
        ids = [reads[pos][1] for pos in at.tolist()]
        files = [f for f, k in files for _ in range(k)]
        bundle = Fast5Batch(files, index, names, ids).as_bundle(reserve)
        return self.enter_fast5_bundle(bundle, names, at, where, table)

    SINGLE_READ_BATCH_MIN = 8

    def prepare_single_read_files(self, reads, where, table, reserve=None):
        """A request that is one SINGLE-read file per read (the reference's classic input): every file opened and its
        read described by one native call on host threads (fast5_file.OpenedFiles) instead of a Python round per file
        (~150 -> ~25 us per read on 8 cores), then the same batch decoders.  Files that cannot be opened, hold another
        read than the one asked for or cannot be described stay with the per-read path, which reports them as the
        reference does.  None: not such a request (a file named twice, a multi-read file among them, a short list)."""
        from .fast5_file import Fast5Batch, OpenedFiles
        n = len(reads)
        names = [key[0] for key in reads]
        if n < self.SINGLE_READ_BATCH_MIN or len(set(names)) != n:
            return None
        opened = OpenedFiles([os.path.join(self.fast5prefix, name) for name in names])
        if opened.multi.any() or (opened.n_reads > 1).any():
            return None
        info = opened.info
        ids = [key[1] for key in reads]
        try:
            asked = np.array([r.encode('ascii') for r in ids], dtype='S64')
        except (UnicodeEncodeError, AttributeError):
            return None
        ok = (opened.rc == 0) & (opened.n_reads == 1) & (info['status'] == 0) & (info['read_id'] == asked) & \
            np.array([len(r) < 64 for r in ids], dtype=bool)
        at = np.nonzero(ok)[0]
        if not len(at):
            return where
        picked = at.tolist()
        batch = Fast5Batch.from_opened(opened, at, [names[k] for k in picked], [ids[k] for k in picked])
        bundle = batch.as_bundle(reserve)
        return self.enter_fast5_bundle(bundle, batch.names, at, where, table)

    def fast5_runs(self, reads):
        """[(Fast5File, name, first read, count)] when the request is stretches of readable multi-read files in
        file order -- a worker batch of a run --, recognised by one list comparison per file; None for anything
        else (single-read files, a shuffled or interleaved request, unreadable reads: the general path)."""
        from .fast5_file import Fast5Error, open_fast5
        runs, pos, n = [], 0, len(reads)
        while pos < n:
            name, read_id = reads[pos]
            try:
                f = open_fast5(os.path.join(self.fast5prefix, name))
            except (OSError, Fast5Error):
                return None
            first = f.index_of(read_id) if f.multi else -1
            if first < 0:
                return None
            count = min(f.n - first, n - pos)
            if reads[pos:pos + count] != f.keys_for(name)[first:first + count] or f.info['status'][first:first + count].any():
                return None
            runs.append((f, name, first, count))
            pos += count
        return runs

    def fast5_call_runs(self, reads):
        """fast5_runs for a worker call, with everything that is made once per FILE made now: the open itself (a
        4 000-read file takes 15-20 ms to walk), its read ids and metadata, its run columns and their plain-run view.
        SignalAnalyzer.process_plain_run calls this OUTSIDE the host phase lock, so that the thread that meets a new
        file first pays for it alone (open_fast5 lets the others that reach the file meanwhile wait for that one walk)."""
        from .fast5_file import file_run_columns
        if self.bundle is not None and any(self.bundle.has_file(name) for name in {key[0] for key in reads}):
            return None                        # (a call that mixes bundle reads and files: the general path sorts it out)
        runs = self.fast5_runs(reads)
        if runs is not None and len(runs) == 1:
            file_run_columns(runs[0][0], runs[0][1]).plain(self.scaler_cfg)
        return runs

    def fast5_run_plan(self, reads, runs=None):
        """The per-call read bundle of a worker call that is stretches of multi-read FAST5 files in file order
        (fast5_runs), laid out from the files' cached metadata with NOTHING decoded yet -- a CallBundle, its sample arena
        from the loader's pool (`call_arenas`: memory a call before it has touched; the caller gives it back); None: not
        such a call.  A run of ONE file (all but the calls that cross a file boundary) is a dozen slices of columns kept
        with the open file (fast5_file.FileRunColumns); several files: a bundle of the call's own."""
        from .fast5_file import Fast5Batch, file_run_columns
        if runs is None:
            runs = self.fast5_call_runs(reads)
        if runs is None:
            return None
        taken = []

        def reserve(n_samples):
            taken.append(self.call_arenas.take(n_samples))
            return taken[0]
        try:
            if len(runs) == 1:
                f, name, i0, count = runs[0]
                cols = file_run_columns(f, name)
                layout = cols.layout(i0, count, reserve)
                plain = cols.plain(self.scaler_cfg)
                if plain is not None:          # the file's columns, the call's text
                    plain = dict(plain, seq_arena=layout['seq_arena'], qual_arena=layout['qual_arena'],
                                 seq_base=int(cols.whole['seq_off'][i0]))
                return CallBundle(runs, layout, plain, i0, taken[0])
            batch = Fast5Batch.from_runs(runs)
            layout = batch.plan(reserve)
            call = CallBundle(runs, layout, None, 0, taken[0])
            call.made = batch.bundle(layout)
            call.plain = call.made.plain_run_columns(self.scaler_cfg)
            return call
        except BaseException:
            if taken:
                self.call_arenas.give(taken[0])
            raise

    def decoding(self):
        """`with loader.decoding() as threads:` around the decode of a worker call's FAST5 reads.  threads: all of the
        host's (None) for a call that is alone in the loader -- its latency --, one -- its own -- when other calls are
        decoding too: the parallelism is then between the calls, and a shared pool that every call wakes up for a
        millisecond of work costs each of them more than it gives."""
        return _Decoding(self)

    def decode_and_run(self, fast, layout, threads, call, offsets, calib):
        """(decoded, rc): the samples and the basecall text of `layout` decoded and -- when all of it arrived -- the
        prepared pxg_process_batch_ex `call` made over them, behind ONE release of the interpreter lock
        (csrc/pxg_pyreport.c decode_and_run: the native functions by address)."""
        from .fast5_file import host_threads
        entry = self.__dict__.get('_decode_entry')
        if entry is None:                 # the two readers' addresses: looked up once
            import ctypes as C
            lib = native.load_text_library()
            entry = self._decode_entry = (C.cast(lib.pxg_h5_load_signals, C.c_void_p).value,
                                          C.cast(lib.pxg_h5_basecall_many, C.c_void_p).value)
        p = layout
        return fast.decode_and_run(
            entry[0], entry[1],
            call.function, int(threads or host_threads()),
            p['handles'], p['index'], p['dst'], p['n_samples'], p['arena'], p['signal_status'],
            p['seq_start'], p['seq_len'], p['seq_arena'], p['qual_arena'], p['move_start'], p['n_moves'], p['move_arena'],
            p['basecall_status'], call.handle, np.ascontiguousarray(offsets, dtype=np.int64),
            np.ascontiguousarray(calib, dtype=native.CALIB_DTYPE), int(call.stage_mask), call.extras, call.records)

    def enter_fast5_bundle(self, bundle, names, at, where, table):
        """Rows for the reads of a Fast5Batch bundle (request positions `at`), the length gate and the reads whose
        samples or text could not be decoded."""
        rows = table.extend_from_bundle(bundle, np.arange(len(at)))
        cfg = self.scaler_cfg      # length gate of load_padded_signal_head (:212-222)
        usable = np.minimum(np.minimum(cfg['length'], table.duration[rows]), table.n_raw[rows])
        short = rows[usable - usable % cfg['stride'] < cfg['min_length']]
        table.halt(short, 'scaler_signal_too_short')
        table.pending[short] = False
        # a read whose signal or basecall text could not be decoded: its own unknown_error
        bad = np.nonzero((bundle.signal_status != 0) | (bundle.basecall_status != 0))[0]
        for k in bad.tolist():
            table.halt(rows[k], 'unknown_error')
            table.pending[rows[k]] = False
            table.error_message[rows[k]] = (
                '({}#{}) FAST5 {} cannot be decoded (native reader code {})'.format(
                    names[k], bundle.read_ids[k], 'signal' if bundle.signal_status[k] else 'basecall',
                    int(bundle.signal_status[k] or bundle.basecall_status[k])))
        where[at] = rows
        return where

    # ---- the GPU pass, in the three steps the session driver overlaps ------------------
    def pack(self, table=None, arena=None, need=None):
        """(rows, arena, offsets, calib) of the reads of `table` that go to the GPU.  With
        `arena` (a page-locked staging buffer, or an object whose reserve(n) returns one) the
        samples are packed in place."""
        t = self.table if table is None else table
        rows = t.live_rows()
        rows = rows[t.pending[rows]]
        offsets = np.zeros(len(rows) + 1, dtype=np.int64)
        np.cumsum(t.n_raw[rows], out=offsets[1:])
        # a run of consecutive bundle reads is already packed: hand out the bundle's own
        # arena (page-locked once by the session) instead of copying 120 KB per read
        bi = t.bundle_index[rows] if len(rows) else np.zeros(0, dtype=np.int64)
        if len(rows) and t.bundle is not None and bi[0] >= 0 and \
                np.array_equal(bi, bi[0] + np.arange(len(rows))):
            t.pending[rows] = False
            # (an int16 view, or the encoded bytes of a compressed bundle: native.EncodedSamples)
            return rows, t.bundle.samples_run(int(bi[0]), int(bi[-1]) + 1), offsets, \
                np.ascontiguousarray(t.calib[rows])
        if hasattr(arena, 'reserve'):
            arena = arena.reserve(int(offsets[-1]))
        if arena is None:
            arena = np.empty(int(offsets[-1]), dtype=np.int16)
        elif len(arena) < offsets[-1]:
            raise ValueError('staging arena too small: {} < {}'.format(len(arena), offsets[-1]))
        for k, i in enumerate(rows.tolist()):
            arena[offsets[k]:offsets[k + 1]] = t.samples_of(i)
            t.raw[i] = None
        t.pending[rows] = False
        return rows, arena[:offsets[-1]], offsets, np.ascontiguousarray(t.calib[rows])

    def run_resident(self, table, rows, offsets):
        """Launch every numeric stage on the batch that is resident on the GPU (uploaded or
        swapped in by the caller) and attach the records to `table`."""
        if len(rows):
            self.ctx.run(self.stage_mask)
            self.collect_resident(table, rows, offsets)

    def collect_resident(self, table, rows, offsets):
        """Second half of run_resident: download the records of the run that was launched on
        the resident batch (waits for it) and attach them to `table`; the chimera window scan
        runs here because it needs the batch's samples still resident."""
        if not len(rows):
            return
        rec = self.ctx.download()
        spikes = self.ctx.download_spikes(rec) if self.stage_mask & native.STAGE_POLYA else None
        self.attach_records(table, rows, rec, spikes)
        if self.dump_adapter:
            # signal[adapter_first : adapter_last + 1] of every read whose adapter was found
            # (signal_analyzer.py:450-453), pooled and scaled where the samples are
            adapter = self.ctx.state_names.index('adapter')
            first, last = rec['seg_first'][:, adapter].astype(np.int64), rec['seg_last'][:, adapter].astype(np.int64)
            count = np.where(first >= 0, last - first + 1, 0)
            table.adapter_dump = self.ctx.pooled_signal(np.maximum(first, 0), count)
        frame = self.unsplit_frames(table, rows, offsets) if self.scan_unsplit or self.dump_events else None
        if self.dump_events:
            # the numeric columns of the dumped event tables (fast5_file.py:209-230,
            # signal_analyzer.py:318), one call per Guppy block stride in the batch
            table.event_frame, table.event_dump = frame, {}
            for stride in np.unique(frame[frame[:, 1] > 0, 2]).tolist():
                sel = (frame[:, 2] == stride) & (frame[:, 1] > 0)
                table.event_dump[int(stride)] = self.ctx.event_table(
                    frame[:, 0], np.where(sel, frame[:, 1], 0), int(stride))
        if self.scan_unsplit:
            for stride in np.unique(frame[frame[:, 1] > 0, 2]).tolist():
                sel = (frame[:, 2] == stride) & (frame[:, 1] > 0)
                self.attach_unsplit(table, rows, sel, self.ctx.unsplit_scan(
                    frame[:, 0], np.where(sel, frame[:, 1], 0), int(stride)))
            own = self.unsplit_event_tables(table, rows)
            if own is not None:          # reads whose basecall brings its own events (albacore)
                n_events, starts, means = own
                self.attach_unsplit(table, rows, n_events > 0, self.ctx.unsplit_scan_events(n_events, starts, means))
                table.own_table_scanned = set(rows[n_events > 0].tolist())

    def attach_records(self, table, rows, rec, spikes, gpu_rows=None):
        """The GPU's records (and spike rows) become the batch table's: scaling-QC verdicts
        (:108-109) and scaling parameters are set here, everything else is judged later.
        `gpu_rows`: the records of `rows` when the table holds only some of the pass's reads."""
        t = table
        t.records = rec
        t.spikes, t.spike_offsets = spikes if spikes is not None else (None, None)
        if gpu_rows is None:
            t.gpu_row[rows] = np.arange(len(rows))
        else:
            t.gpu_row[rows] = gpu_rows
            rec = rec[gpu_rows]
        qc_failed = rec['status'] == native.STATUS_CODE['scaling_qc_fail']
        t.halt(rows[qc_failed], 'scaling_qc_fail')
        good = rows[~qc_failed]
        t.scale_shift[good, 0], t.scale_shift[good, 1] = rec['scale'][~qc_failed], rec['shift'][~qc_failed]
        t.has_scaling[good] = True

    def attach_unsplit(self, table, rows, sel, scanned):
        """Candidate lists of one window scan (reads `sel` of the batch took part)."""
        iv, cnt, start = scanned
        picked = np.nonzero(sel)[0]
        table.unsplit_count[rows[picked]] = cnt[picked]
        for k in picked[cnt[picked] > 0].tolist():
            table.unsplit[rows[k]] = iv[start[k]:start[k + 1]].tolist()

    def pin_bundle(self):
        """Page-lock the bundle's sample arena (or its encoded bytes + chunk records) once, so
        every batch of consecutive bundle reads goes to the GPU as a DMA transfer straight from
        the bundle; stays pinned for the life of the worker (unpin_bundle)."""
        if self.bundle is None or self._pinned:
            return
        with self._stage_lock:           # worker calls may start on several threads at once
            if self._pinned:
                return
            d = self.bundle.d
            pinned = []
            for key in (('arena_z', 'z_chunks') if self.bundle.compressed else ('arena',)):
                if d[key].nbytes:
                    # (a small array comes from the malloc heap and shares pages with its neighbours: page-lock a
                    #  copy that has its pages to itself -- native.pinnable)
                    d[key] = a = native.pinnable(d[key])
                    self.ctx.pin(a)
                    pinned.append(a)
            self._pinned = pinned

    def unpin_bundle(self):
        self.call_arenas.release()
        pinned, self._pinned = self._pinned, []
        for a in pinned:
            self.ctx.unpin(a)

    def fit_scalers(self, table=None):
        """The GPU pass over every read of the table that is still live: scaler network +
        QC and, behind the same copy, every other numeric stage.

        The reference keeps `parallel` worker calls in flight (pipeline.py:96,204-205); here
        calls from several threads of ONE process share the GPU context and overlap on it: a
        call copies its samples into the spare input slot on the copy stream (pxg_batch_stage)
        while the previous call's kernels run on the resident batch, becomes resident
        (pxg_batch_swap) once that call has downloaded its records, launches, and builds its
        result dicts while the next call computes.  Two locks, always taken in this order."""
        t = self.table if table is None else table
        rows, arena, offsets, calib = self.pack(t)
        if not len(rows):
            return
        self.pin_bundle()
        polya = bool(self.stage_mask & native.STAGE_POLYA)
        if hasattr(self.ctx, 'process_batch_ex') and not (self.dump_adapter or self.dump_events):
            # one native call per worker batch: stage / swap / run / downloads happen inside it
            # with the GIL released (include/pxg.h, pxg_process_batch_ex)
            frame = self.unsplit_frames(t, rows, offsets) if self.scan_unsplit else None
            strides = np.unique(frame[frame[:, 1] > 0, 2]).tolist() if frame is not None else []
            scan = None
            if len(strides) == 1:          # (several block strides in one batch: rare, scanned below)
                sel = (frame[:, 2] == strides[0]) & (frame[:, 1] > 0)
                scan = (frame[:, 0], np.where(sel, frame[:, 1], 0), int(strides[0]))
            own_tables = self.scan_unsplit and self.has_own_event_tables(t, rows)
            if len(strides) <= 1 and not own_tables:
                got = self.ctx.process_batch_ex(arena, offsets, calib, self.stage_mask, unsplit=scan,
                                                want_spikes=polya)
                self.attach_records(t, rows, got['records'], got.get('spikes'))
                if scan is not None:
                    self.attach_unsplit(t, rows, sel, got['unsplit'])
                return
        # the same steps from Python (a context double in the CPU tests, dump options, a batch that mixes
        # Guppy block strides or carries albacore tables) under the SAME two locks the one-call form takes
        # inside the library (pxg_ctx_lock: calls of both forms may be in flight on other threads)
        ctx = self.ctx
        native_locks = hasattr(ctx, 'lock')
        lock = ctx.lock if native_locks else (lambda w: (self._run_lock if w else self._stage_lock).acquire())
        unlock = ctx.unlock if native_locks else (lambda w: (self._run_lock if w else self._stage_lock).release())
        lock(0)
        try:
            limit = {}
            if hasattr(ctx, 'prefix_limit_for'):              # (dumps and scans walk whole reads)
                whole = bool(self.scan_unsplit or self.dump_events)
                limit = {'prefix_limit': ctx.prefix_limit_for(self.stage_mask, whole)}
            if isinstance(arena, native.EncodedSamples):      # compressed bundle: decoded on the GPU
                ctx.stage_z(arena, offsets, calib, **limit)
            else:
                ctx.stage(arena, offsets, calib, **limit)
            lock(1)                                           # the previous call has its records
            try:
                ctx.swap()
            except BaseException:
                unlock(1)
                raise
        finally:
            unlock(0)
        try:
            self.run_resident(t, rows, offsets)
        finally:
            unlock(1)

    def records_of_run(self, arena, offsets, calib, scan=None):
        """The GPU pass of fit_scalers for a batch that needs nothing but its records, -- with the poly(A) stage -- its
        spike rows and -- `scan` = (first sample, blocks, block stride) -- the candidates of the chimera window scan (no
        dumps): {'records', 'spikes': (rows, offsets), 'unsplit': (intervals, count, start)} from the one native call,
        or -- a context double of the CPU tests, without a scan -- from the same steps under the same two locks."""
        ctx = self.ctx
        polya = bool(self.stage_mask & native.STAGE_POLYA)
        if hasattr(ctx, 'process_batch_ex'):
            return ctx.process_batch_ex(arena, offsets, calib, self.stage_mask, unsplit=scan, want_spikes=polya)
        if scan is not None:
            raise ValueError('a scan needs the one-call form of the context')
        native_locks = hasattr(ctx, 'lock')
        lock = ctx.lock if native_locks else (lambda w: (self._run_lock if w else self._stage_lock).acquire())
        unlock = ctx.unlock if native_locks else (lambda w: (self._run_lock if w else self._stage_lock).release())
        lock(0)
        try:
            limit = {}
            if hasattr(ctx, 'prefix_limit_for'):
                limit = {'prefix_limit': ctx.prefix_limit_for(self.stage_mask, False)}
            if isinstance(arena, native.EncodedSamples):
                ctx.stage_z(arena, offsets, calib, **limit)
            else:
                ctx.stage(arena, offsets, calib, **limit)
            lock(1)
            try:
                ctx.swap()
            except BaseException:
                unlock(1)
                raise
        finally:
            unlock(0)
        try:
            ctx.run(self.stage_mask)
            rec = ctx.download()
            return {'records': rec, 'spikes': ctx.download_spikes(rec) if polya else None}
        finally:
            unlock(1)

    def has_own_event_tables(self, table, rows):
        """Does any read of the batch carry an albacore Events table (its own event boundaries)?"""
        t = table
        if t.bundle is not None:
            bi = t.bundle_index[rows]
            if (t.bundle.d['bc_table'][np.where(bi >= 0, bi, 0)][bi >= 0] == 3).any():
                return True
            if (bi >= 0).all():
                return False
        for k in np.nonzero(t.bundle_index[rows] < 0)[0].tolist() if t.bundle is not None else range(len(rows)):
            try:
                bc = t.source_of(rows[k]).get_basecall()
            except Exception:
                continue
            if bc is not None and bc.get('table') == 'albacore':
                return True
        return False

    def unsplit_event_tables(self, table, rows):
        """(n_events [n], start arena int64, mean arena float32) of the batch's reads whose basecall is an
        albacore Events table the window scan can take as it is: a signed-integer, ascending `start'
        column and a float32 `mean' column.  Every other such table is left out here and fails (or is
        refused) per read, where the reference fails it (SignalAnalysis.detect_unsplit_read).  None when
        the batch has no such read."""
        t = table
        n = len(rows)
        n_events = np.zeros(n, dtype=np.int64)
        starts, means = [], []
        kinds = None
        if t.bundle is not None:
            bi = t.bundle_index[rows]
            kinds = np.where(bi >= 0, t.bundle.d['bc_table'][np.where(bi >= 0, bi, 0)], -1)
        for k in range(n):
            if kinds is not None and kinds[k] >= 0 and kinds[k] != 3:
                continue
            try:
                bc = t.source_of(rows[k]).get_basecall()
            except Exception:
                continue
            ev = bc.get('events') if bc is not None and bc.get('table') == 'albacore' else None
            if ev is None or 'start' not in ev or 'mean' not in ev:
                continue
            st, mean = np.asarray(ev['start']), np.asarray(ev['mean'])
            if st.dtype.kind != 'i' or mean.dtype != np.float32 or not len(st) or st[0] < 0 or (np.diff(st) < 0).any():
                continue
            n_events[k] = len(st)
            starts.append(st.astype(np.int64))
            means.append(mean)
        if not starts:
            return None
        return n_events, np.concatenate(starts), np.concatenate(means)

    def unsplit_frames(self, table, rows, offsets):
        """[n, 3] (first sample, Guppy blocks, block stride) of the batch's reads for the a18 +
        a19 stages (Guppy block means of every basecalled read and the windowed Viterbi scan,
        signal_analyzer.py:366-418, both on the GPU); blocks = 0 leaves a read out.  Reads whose
        event frame cannot be built are left out here; load_events raises for them later, per
        read.  Host-only: computed BEFORE the batch goes to the GPU."""
        t = table
        n = len(rows)
        frame = np.zeros((n, 3), dtype=np.int64)        # first sample, blocks, block stride
        n_raw = np.diff(offsets)
        per_read = np.ones(n, dtype=bool)
        if t.bundle is not None:                        # bundle rows: the frame from the columns
            d, bi = t.bundle.d, t.bundle_index[rows]
            b = np.where(bi >= 0, bi, 0)
            first, stride, moves = d['bc_first_sample'][b], d['bc_block_stride'][b].astype(np.int64), d['bc_n_moves'][b]
            covered = np.maximum(np.minimum(first + stride * moves, n_raw) - first, 0)
            kind = d['bc_table'][b]
            fits = (bi >= 0) & d['bc_present'][b] & ((kind == 1) | (kind == 2)) & (moves >= 0) & \
                (stride > 0) & (-(-covered // np.maximum(stride, 1)) == moves)
            frame[fits] = np.stack([first, moves, stride], axis=1)[fits]
            per_read = bi < 0                           # everything else in a bundle has no usable frame
        for k in np.nonzero(per_read)[0].tolist():
            try:
                frame[k] = NanoporeRead(t, rows[k]).guppy_event_geometry(n_raw[k])
            except Exception:
                pass
        return frame
