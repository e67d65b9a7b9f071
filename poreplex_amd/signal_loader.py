"""Per-read record and batch loader (reference: poreplex/signal_loader.py).

``NanoporeRead`` keeps the reference's attribute/report surface
(signal_loader.py:112-198).  ``SignalLoader`` gathers the int16 DAQ samples of
every read of a worker batch into ONE packed arena and runs all numeric stages
in a single GPU call (``fit_scalers``), where the reference makes one Keras
``predict`` per batch and then loops over reads in Python
(signal_loader.py:89-109, signal_analyzer.py:107-123).
"""
import os

import numpy as np

from . import native
from .fast5_file import ReadBundle, open_read

__all__ = ['SignalLoader', 'NanoporeRead', 'SignalAnalysisError']


class SignalAnalysisError(Exception):
    pass


class SignalLoader:

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
        self.batch_reads = []
        self.stage_mask = native.STAGE_ALL_DEMUX
        self.scan_unsplit = False      # --filter-chimera: also run the a19 window scan

    def clear(self):
        del self.batch_reads[:]

    def exists(self, filename):
        if self.bundle is not None and self.bundle.has_file(filename):
            return True
        return os.path.exists(os.path.join(self.fast5prefix, filename))

    def prepare_loading(self, filename, read_id):
        npread = NanoporeRead(filename, self.fast5prefix, read_id, self.bundle)
        # signal_loader.py:212-222: the length gate of load_padded_signal_head
        npread.check_signal_head(self.scaler_cfg['length'], self.scaler_cfg['stride'],
                                 self.scaler_cfg['min_length'])
        if not npread.is_stopped():
            self.batch_reads.append(npread)
        return npread

    def fit_scalers(self):
        """One GPU pass over every loaded read: scaler net + QC, pooling,
        Viterbi segmentation, barcode window + classifier (+ poly(A))."""
        if not self.batch_reads:
            return
        sigs = [r.raw for r in self.batch_reads]
        arena, offsets = native.pack_reads(sigs)
        calib = np.zeros(len(sigs), dtype=native.CALIB_DTYPE)
        for i, r in enumerate(self.batch_reads):
            calib[i] = (r.fast5.range, r.fast5.digitization, r.fast5.offset, r.fast5.sampling_rate)
        self.ctx.upload(arena, offsets, calib)
        self.ctx.run(self.stage_mask)
        records = self.ctx.download()
        spikes = self.ctx.download_spikes() if self.stage_mask & native.STAGE_POLYA else None
        qc_fail = native.STATUS_CODE['scaling_qc_fail']
        for i, r in enumerate(self.batch_reads):
            r.native = records[i]
            r.native_spikes = None if spikes is None else spikes[i]
            if records[i]['status'] == qc_fail:                 # signal_loader.py:108-109
                r.set_status('scaling_qc_fail', stop=True)
            else:
                r.set_scaling_params(np.array([records[i]['scale'], records[i]['shift']],
                                              dtype=np.float32))
        if self.scan_unsplit:
            self.scan_unsplit_candidates(offsets)
        for r in self.batch_reads:
            r.raw = None

    def scan_unsplit_candidates(self, offsets):
        """a18+a19 numeric part for the resident batch: Guppy block means of
        every basecalled read and the windowed Viterbi scan, on the GPU
        (signal_analyzer.py:366-418).  Reads whose event table cannot be built
        are skipped here; SignalAnalysis.load_events raises for them."""
        n = len(self.batch_reads)
        first = np.zeros(n, dtype=np.int64)
        blocks = np.zeros(n, dtype=np.int64)
        strides = np.zeros(n, dtype=np.int64)
        for i, r in enumerate(self.batch_reads):
            try:
                table = r.guppy_event_geometry(int(offsets[i + 1] - offsets[i]))
            except Exception:
                continue
            first[i], blocks[i], strides[i] = table
        for stride in sorted(set(strides[blocks > 0].tolist())):
            sel = (strides == stride) & (blocks > 0)
            iv, cnt = self.ctx.unsplit_scan(first, np.where(sel, blocks, 0), int(stride))
            for i in np.nonzero(sel)[0]:
                r = self.batch_reads[i]
                r.native_unsplit_count = int(cnt[i])
                r.native_unsplit = iv[i, :min(int(cnt[i]), iv.shape[1])].tolist()


class NanoporeRead:

    fast5 = error_message = None
    sequence_length = mean_qscore = num_events = 0
    sequence = scaling_params = label = barcode = polya = None
    barcode_bestguess = barcode_quality = None
    native = native_spikes = raw = None
    native_unsplit = None
    native_unsplit_count = 0

    def __init__(self, filename, srcdir, read_id, bundle=None):
        self.fullpath = os.path.join(srcdir, filename)
        self.filename = filename
        self.read_id = read_id
        self.status = 'okay'
        self.stopped = False
        self.load(bundle)

    def set_status(self, newstatus, stop=False):
        self.status = newstatus
        self.stopped = self.stopped or stop

    def set_error(self, status, error_message):
        self.status = status
        self.error_message = error_message

    def set_scaling_params(self, params):
        self.scaling_params = params

    def set_label(self, newlabel):
        self.label = newlabel

    def set_barcode(self, newbarcode, guess, quality):
        self.barcode = newbarcode
        self.barcode_bestguess = guess
        self.barcode_quality = quality

    def set_adapter_trimming_length(self, newlength):
        if self.sequence is None:
            raise Exception('Sequence is not set.')
        self.sequence = self.sequence[:2] + (newlength,)

    def set_polya_tail(self, polya_info):
        self.polya = polya_info

    def is_stopped(self):
        return self.stopped

    def close(self):
        self.raw = None
        if self.fast5 is not None:
            self.fast5.close()

    def report(self):
        """Result dict, keys and order as signal_loader.py:165-198."""
        rep = {'filename': self.filename, 'read_id': self.read_id, 'status': self.status}
        if self.fast5 is not None:
            rep.update({
                'channel': self.fast5.channel_number,
                'start_time': round(self.fast5.start_time / self.fast5.sampling_rate, 3),
                'run_id': self.fast5.run_id,
                'sample_id': self.fast5.sample_id,
                'duration': self.fast5.duration,
                'num_events': self.num_events,
                'sequence_length': self.sequence_length,
                'mean_qscore': self.mean_qscore,
            })
        if self.sequence is not None:
            rep['sequence'] = self.sequence
        if self.error_message:
            rep['error_message'] = self.error_message
        if self.label is not None:
            rep['label'] = self.label
        if self.barcode is not None:
            rep['barcode'] = self.barcode
            rep['barcode_guess'] = self.barcode_bestguess
            rep['barcode_score'] = self.barcode_quality
        if self.polya is not None:
            rep['polya'] = self.polya
        return rep

    def load(self, bundle=None):
        # The reference marks an unreadable file 'irregular_fast5' here
        # (signal_loader.py:200-207) and then trips over fast5 == None in
        # load_padded_signal_head, so the caller reports 'unknown_error'
        # (SURVEY App. C / golden batch0): keep that observable behaviour by
        # letting the exception reach SignalAnalyzer.process.
        self.fast5 = open_read(self.fullpath, self.filename, self.read_id, bundle)
        self.sampling_rate = self.fast5.sampling_rate

    def check_signal_head(self, length_limit, stride, min_length):
        self.raw = np.ascontiguousarray(self.fast5.get_raw_int16(), dtype=np.int16)
        sigload_length = min(length_limit, self.fast5.duration, len(self.raw))
        sigload_length -= sigload_length % stride
        if sigload_length < min_length:
            self.set_status('scaler_signal_too_short', stop=True)
            self.raw = None

    def load_fast5_events(self):
        """Basecall summary of the FAST5 (signal_loader.py:266-279); the event
        table itself is only materialised by the stages that consume it."""
        if self.fast5 is None:
            raise Exception('Fast5 must be open for getting events.')
        bcall = self.fast5.get_basecall()
        if bcall is None:
            raise SignalAnalysisError('not_basecalled')
        self.sequence_length = bcall['sequence_length']
        self.mean_qscore = bcall['mean_qscore']
        self.num_events = bcall['num_events']
        self.sequence = bcall['sequence'], bcall['qstring'], 0
        return bcall

    def guppy_event_geometry(self, n_raw=None):
        """(first_sample, n_blocks, block_stride) of the Move-table event
        frame, with the size rule of convert_events_guppy (fast5_file.py:
        210-223): the raw slice, NaN-padded to whole blocks, must hold exactly
        one block per move."""
        bcall = self.fast5.get_basecall()
        if bcall is None:
            raise SignalAnalysisError('not_basecalled')
        if bcall.get('move') is None:
            raise Exception("Neither `Events' or `Move' table found in the basecall.")
        if n_raw is None:
            n_raw = len(self.fast5.get_raw_int16())
        first, stride = int(bcall['first_sample_template']), int(bcall['block_stride'])
        n_blocks = len(bcall['move'])
        length = max(min(first + stride * n_blocks, n_raw) - first, 0)
        padded = length + (stride - length % stride if length % stride else 0)
        if padded // stride != n_blocks:
            raise Exception('Numbers of events and raw data strides does not match.')
        return first, n_blocks, stride
