"""Result sinks of the per-read processor: what the pipeline does with
the list of result dicts `process_batch` returns.

  setup_output_name_mapping   poreplex/commandline.py:137-159
  SequencingSummaryWriter     poreplex/io.py:120-184   (sequencing_summary.txt)
  FinalSummaryTracker         poreplex/io.py:236-332   (the end-of-run table)
  FASTQWriter                 poreplex/io.py:40-74     (routing by (label, barcode))

Same constructor arguments, method names and -- byte for byte -- the same text as the
reference's classes on the same result dicts (tests/golden/sinks.json was produced by
running the real ones).  Differences by construction: no pandas (the summary table
is ordered with plain sorts), gzip instead of pysam's BGZF container for FASTQ, and
`FinalSummaryTracker.feed_counts`, which takes the all-reduced
[label x barcode x status] table of `poreplex_amd.distributed.reduce_counts` so that
rank 0 of a multi-GPU run prints the whole run's summary.  FAST5 / nanopolish /
BAM writers stay out of scope (file plumbing, no signal path).
"""
import gzip
import logging
import os

import numpy as np
from collections import defaultdict
from threading import Lock

from .native import STATUS_NAMES

__all__ = ['setup_output_name_mapping', 'SequencingSummaryWriter', 'FinalSummaryTracker',
           'FASTQWriter', 'OUTPUT_NAME_PASSED', 'OUTPUT_NAME_FAILED', 'OUTPUT_NAME_ARTIFACT',
           'OUTPUT_NAME_UNDETERMINED', 'OUTPUT_NAME_BARCODES', 'OUTPUT_NAME_BARCODING_OFF']

# poreplex/__init__.py:32-38
OUTPUT_NAME_PASSED = 'pass'
OUTPUT_NAME_FAILED = 'fail'
OUTPUT_NAME_ARTIFACT = 'artifact'
OUTPUT_NAME_UNDETERMINED = 'undetermined'
OUTPUT_NAME_BARCODES = 'BC{n}'
OUTPUT_NAME_BARCODING_OFF = '-'


def setup_output_name_mapping(config):
    """-> (label_names, barcode_names, layout) with layout[(label, barcode)] = relative
    output name (commandline.py:137-159)."""
    label_names = {'fail': OUTPUT_NAME_FAILED, 'pass': OUTPUT_NAME_PASSED}
    if config['filter_unsplit_reads']:
        label_names['artifact'] = OUTPUT_NAME_ARTIFACT
    if not config['barcoding']:
        return (label_names, {None: OUTPUT_NAME_BARCODING_OFF},
                {(label, None): name for label, name in label_names.items()})
    barcode_names = {None: OUTPUT_NAME_UNDETERMINED}
    for i in range(config['demultiplexing']['number_of_barcodes']):
        barcode_names[i] = OUTPUT_NAME_BARCODES.format(n=i + 1)
    layout = {(label, bc): os.path.join(lname, bname)
              for label, lname in label_names.items() for bc, bname in barcode_names.items()}
    return label_names, barcode_names, layout


def _ensure_parent(path):
    parent = os.path.dirname(path)
    if parent and not os.path.isdir(parent):
        os.makedirs(parent, exist_ok=True)


class SequencingSummaryWriter:
    """One tab-separated row per LABELLED result (reads that never got past loading have
    no label and are skipped, io.py:166-168)."""

    SUMMARY_OUTPUT_FIELDS = ['filename', 'read_id', 'run_id', 'channel', 'start_time', 'duration',
                             'num_events', 'sequence_length', 'mean_qscore', 'sample_id', 'status',
                             'label']

    def __init__(self, config, output_dir, label_mapping, barcode_mapping, suffix='', header=True):
        """`suffix` / `header`: a multi-GPU session writes one part file per rank (only rank
        0's carries the header) and stitches them together in rank order."""
        self.lock = Lock()
        self.label_mapping = label_mapping
        self.barcode_mapping = barcode_mapping if config['barcoding'] else None
        self.file = open(os.path.join(output_dir, 'sequencing_summary.txt' + suffix), 'w')
        self.polya_enabled = bool(config['measure_polya'])
        self.fast5_layout = bool(config['fast5_output'])
        self.output_fields = list(self.SUMMARY_OUTPUT_FIELDS)
        if self.barcode_mapping is not None:
            self.output_fields += ['barcode', 'barcode_score']
        if self.polya_enabled:
            self.output_fields.append('polya_dwell')
        if header:
            self.file.write('\t'.join(self.output_fields) + '\n')
        self._row_scratch = []

    def close(self):
        self.file.close()

    def format_filename(self, row):
        # row['label'] is already the output name here, like in the reference's closures
        if not self.fast5_layout:
            return row['filename']
        if self.barcode_mapping is not None:
            return os.path.join('fast5', row['label'], self.barcode_mapping[row.get('barcode')],
                                row['filename'])
        return os.path.join('fast5', row['label'], row['filename'])

    def write_results(self, results):
        with self.lock:
            for entry in results:
                if 'label' not in entry:
                    continue
                row = dict(entry)
                row['label'] = self.label_mapping[entry['label']]
                row['filename'] = self.format_filename(row)
                if self.barcode_mapping is not None:
                    row['barcode'] = self.barcode_mapping[entry.get('barcode')]
                    row['barcode_score'] = entry.get('barcode_score', 0)
                if self.polya_enabled:
                    row['polya_dwell'] = (format(entry['polya']['dwell_time'], '.4f')
                                          if 'polya' in entry else '')
                self.file.write('\t'.join(str(row[f]) for f in self.output_fields) + '\n')


    def write_table_rows(self, table, rows):
        """The rows of `rows` (all labelled, all from table.bundle) through the library's
        formatter (pxg_summary_rows): same bytes as write_columns(summary_columns(...)), without
        a Python object per field.  False if it does not apply (the caller then takes the
        Python path): fast5 output layout, reads that are not in a bundle, non-ASCII names."""
        from . import native
        bundle = table.bundle
        idx = np.asarray(rows, dtype=np.int64)
        if self.fast5_layout or bundle is None or not len(idx) or (table.bundle_index[idx] < 0).any():
            return False
        d = bundle.d
        labels = [self.label_mapping.get(name, '') for name in ('pass', 'fail', 'artifact')]
        bc = {}
        if self.barcode_mapping is not None:
            called = table.has_barcode[idx]
            n_bc = max([k for k in self.barcode_mapping if k is not None], default=-1) + 1
            bc = dict(barcode=np.where(called, table.barcode[idx], -1),
                      barcode_score=np.where(called, table.barcode_phred[idx], 0),
                      barcode_names=[self.barcode_mapping[None]] + [self.barcode_mapping[k] for k in range(n_bc)])
        pa = {}
        if self.polya_enabled:
            lazy = table.polya_lazy[idx]
            dwell = table.polya_dwell_time[idx].copy()
            has = lazy.copy()
            for k in np.nonzero(~lazy)[0].tolist():          # tails set through set_polya_tail (dicts)
                p = table.polya[int(idx[k])]
                if p is not None:
                    has[k], dwell[k] = True, p['dwell_time']
            pa = dict(has_polya=has, polya_dwell=dwell)
        text = native.summary_rows(
            [d['filename'], d['read_id'], d['run_id'], d['channel_number'], d['sample_id']],
            table.bundle_index[idx], table.start_time[idx], table.sampling_rate[idx], table.duration[idx],
            table.has_summary[idx], table.num_events[idx], table.sequence_length[idx], table.mean_qscore[idx],
            table.status[idx], native.STATUS_NAMES, table.label[idx], labels, scratch=self._row_scratch, **bc, **pa)
        if text is None:
            return False
        with self.lock:
            self.file.write(text.decode('ascii'))
        return True

    def write_columns(self, cols):
        """write_results for rows that only exist as columns (signal_loader.summary_columns):
        same text, no per-read dicts.  Every row must carry a label."""
        n = len(cols['read_id'])
        label = [self.label_mapping[v] for v in cols['label']]
        bcname = [self.barcode_mapping[b] for b in cols['barcode']] \
            if self.barcode_mapping is not None else None
        if not self.fast5_layout:
            filename = cols['filename']
        elif bcname is not None:
            filename = [os.path.join('fast5', a, b, f) for a, b, f in zip(label, bcname, cols['filename'])]
        else:
            filename = [os.path.join('fast5', a, f) for a, f in zip(label, cols['filename'])]
        out = dict(cols, label=label, filename=filename)
        if bcname is not None:
            out['barcode'] = bcname
        if self.polya_enabled:
            out['polya_dwell'] = [format(d, '.4f') if d is not None else '' for d in cols['polya_dwell_time']]
        fields = [col if (n and type(col[0]) is str and f not in ('barcode_score',)) else list(map(str, col))
                  for f, col in ((f, out[f]) for f in self.output_fields)]
        text = ''.join('\t'.join(row) + '\n' for row in zip(*fields)) if n else ''
        with self.lock:
            self.file.write(text)


class FinalSummaryTracker:
    """Counts per (label, barcode, status) and the end-of-run table."""

    REPORTING_ORDER = ['pass', 'artifact', 'fail']
    FRIENDLY_LABELS = {'pass': 'Successfully processed', 'fail': 'Processing failed',
                       'artifact': 'Possible artifact'}
    # label / status / the wording of the end-of-run table (io.py:245-260; output text)
    FRIENDLY_STATUS = {}
    for _row in ("fail|scaler_signal_too_short|Signal is too short",
                 "fail|sequence_too_short|Sequence is too short",
                 "fail|irregular_fast5|Invalid FAST5 format",
                 "fail|basecall_table_incomplete|Basecall table does not match",
                 "fail|adapter_not_detected|3' Adapter could not be located",
                 "fail|not_basecalled|No albacore basecall data found",
                 "fail|scaling_qc_fail|Signal scaling QC failed",
                 "fail|disappeared|File is moved to other location",
                 "fail|unknown_error|File could not be opened due to unknown error",
                 "artifact|unsplit_read|Two or more molecules found within a read"):
        _label, _status, _text = _row.split('|')
        FRIENDLY_STATUS.setdefault(_label, {})[_status] = _text
    del _row, _label, _status, _text
    LABEL_FORMAT = '{:49s} '
    LABEL_BULLET = ' - '
    MINIMUM_COLUMN_WIDTH = 3

    def __init__(self, label_names, barcode_names):
        self.label_names, self.barcode_names = label_names, barcode_names
        self.counts = defaultdict(int)          # (label, barcode, status) -> reads
        self.label_reporting_order = self.REPORTING_ORDER
        self.barcode_reporting_order = sorted(n for n in barcode_names if n is not None) + [None]

    def feed_results(self, results):
        for entry in results:
            self.counts[entry.get('label', 'fail'), entry.get('barcode', None), entry['status']] += 1

    def feed_counts(self, table, label_order=('pass', 'fail', 'artifact')):
        """Add an int64 [label, barcode slot, status] table (distributed.count_table /
        reduce_counts: slot 0 = undetermined, slot k = barcode k-1)."""
        for li, label in enumerate(label_order):
            for slot in range(table.shape[1]):
                for si in range(table.shape[2]):
                    n = int(table[li, slot, si])
                    if n:
                        self.counts[label, None if slot == 0 else slot - 1, STATUS_NAMES[si]] += n

    def print_results(self, file):
        if hasattr(file, 'write'):
            def emit(line):
                file.write(line + '\n')
        else:
            logger = logging.getLogger('poreplex')
            emit = logger.error

        emit('==== Result Summary ====')
        width = max(self.MINIMUM_COLUMN_WIDTH, len(format(max(self.counts.values()), 'd')))
        title = '{{:{}s}} '.format(width)
        number = '{{:{}d}} '.format(width)
        if len(self.barcode_names) > 1:
            emit(self.LABEL_FORMAT.format('') +
                 ''.join(title.format(self.barcode_names[bc]) for bc in self.barcode_reporting_order))

        # rows in feeding order, stably sorted by (label rank, count descending); one output
        # line per (label, status) in order of first appearance -- what the reference gets from
        # DataFrame.sort_values(...).groupby(..., sort=False)
        rows = [(label, bc, status, n) for (label, bc, status), n in self.counts.items()]
        rows.sort(key=lambda r: (self.label_reporting_order.index(r[0]), -r[3]))
        groups = {}
        for label, bc, status, n in rows:
            groups.setdefault((label, status), {})[bc] = n
        current = None
        for (label, status), by_barcode in groups.items():
            line_title = None
            if label != current:
                current = label
                if label in self.FRIENDLY_STATUS:
                    emit(self.LABEL_FORMAT.format(self.FRIENDLY_LABELS[label]))
                else:
                    line_title = self.FRIENDLY_LABELS[label]
            if line_title is None:
                line_title = self.LABEL_BULLET + self.FRIENDLY_STATUS[label][status]
            emit(self.LABEL_FORMAT.format(line_title) +
                 ''.join(number.format(by_barcode.get(bc, 0)) for bc in self.barcode_reporting_order))
        emit('')


class FASTQWriter:
    """<output_dir>/fastq/<layout name>.fastq.gz per (label, barcode); sequences lose their
    adapter_length trailing bases (io.py:63-74)."""

    def __init__(self, output_dir, output_layout, suffix=''):
        self.output_dir, self.output_layout, self.suffix = output_dir, output_layout, suffix
        self.lock, self.streams = Lock(), {}
        for key, name in output_layout.items():
            path = self.get_output_path(name)
            _ensure_parent(path)
            self.streams[key] = gzip.open(path, 'wb')

    def get_output_path(self, name):
        return os.path.join(self.output_dir, 'fastq', name + '.fastq.gz' + self.suffix)

    def close(self):
        for stream in self.streams.values():
            stream.close()

    def write_sequences(self, procresult):
        with self.lock:
            for entry in procresult:
                if entry.get('sequence') is None:
                    continue
                seq, qual, adapter_length = entry['sequence']
                if adapter_length > 0:
                    seq, qual = seq[:-adapter_length], qual[:-adapter_length]
                record = '@{}\n{}\n+\n{}\n'.format(entry['read_id'], seq, qual)
                self.streams[entry['label'], entry.get('barcode')].write(record.encode('ascii'))
