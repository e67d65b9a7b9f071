"""Per-read processor: the reference's operator surface over the GPU hot path.

Drop-in for poreplex/signal_analyzer.py: ``process_batch(batchid, reads,
config)`` (picklable, :46-58), ``SignalAnalyzer(config, batchid)`` context
manager with ``.process(reads)`` (:61-134) and ``SignalAnalysis`` (:214-466).
Same signatures, result-dict schema, status/label vocabulary and result order;
every numeric stage between int16 DAQ samples and the per-read record runs in
HIP behind the C ABI (include/pxg.h).  There is no CPU fallback: a missing
library or GPU surfaces as the ``(-1, msg, traceback)`` tuple the pipeline
treats as fatal (pipeline.py:207-213).
"""
import os
import sys
import traceback
from io import StringIO
from weakref import proxy

import numpy as np

from . import native
from .signal_loader import SignalAnalysisError
from .utils import union_intervals  # noqa: F401  (re-exported like the reference)
from .worker_persistence import WorkerPersistenceStorage

__all__ = ['SignalAnalyzer', 'SignalAnalysis', 'process_batch']


# This function must be picklable.
def process_batch(batchid, reads, config):
    try:
        with SignalAnalyzer(config, batchid) as analyzer:
            return analyzer.process(reads)
    except Exception as exc:
        exc_type, exc_obj, exc_tb = sys.exc_info()
        filename = os.path.split(exc_tb.tb_frame.f_code.co_filename)[-1]
        errorf = StringIO()
        traceback.print_exc(file=errorf)
        return (-1, '[{filename}:{lineno}] Unhandled exception {name}: {msg}'.format(
            filename=filename, lineno=exc_tb.tb_lineno,
            name=type(exc).__name__, msg=str(exc)), errorf.getvalue())


class SignalAnalyzer:

    def __init__(self, config, batchid):
        WorkerPersistenceStorage(config).retrieve_objects(self)
        self.config = config
        self.inputdir = config['inputdir']
        self.outputdir = config['outputdir']
        self.batchid = batchid
        self.formatted_batchid = format(batchid, '08d')
        if config.get('dump_adapter_signals') or config.get('dump_basecalls'):
            raise NotImplementedError('HDF5 dump outputs are outside the hot path')
        mask = native.STAGE_SCALER | native.STAGE_SEGMENT
        if config['barcoding']:
            mask |= native.STAGE_BARCODE
        if config['measure_polya']:
            mask |= native.STAGE_POLYA
        self.loader.stage_mask = mask
        self.loader.scan_unsplit = bool(config.get('filter_unsplit_reads'))

    def process(self, reads):
        results, loaded = [], []
        nextprocs = []
        prepare_loading = self.loader.prepare_loading
        for f5file, read_id in reads:
            if not self.loader.exists(f5file):
                results.append({'filename': f5file, 'status': 'disappeared'})
                continue
            try:
                npread = prepare_loading(f5file, read_id)
                if npread.is_stopped():
                    results.append(npread.report())
                else:
                    nextprocs.append(SignalAnalysis(npread, self))
                    loaded.append(npread)
            except Exception as exc:
                results.append(self.pack_unhandled_exception(f5file, read_id, exc, sys.exc_info()))

        # scaling parameters -- and, in the same GPU pass, every other numeric stage
        self.loader.fit_scalers()

        for siganal in nextprocs:
            try:
                if not siganal.is_stopped():
                    siganal.process()
            except Exception as exc:
                error = self.pack_unhandled_exception(siganal.npread.filename,
                                                      siganal.npread.read_id, exc, sys.exc_info())
                siganal.set_error(error)
            finally:
                siganal.clear_cache()

        if self.config['barcoding']:
            self.demuxer.predict()

        for npread in loaded:
            results.append(npread.report())
        return results

    def pack_unhandled_exception(self, f5filename, read_id, exc, excinfo):
        exc_type, exc_obj, exc_tb = excinfo
        srcfilename = os.path.split(exc_tb.tb_frame.f_code.co_filename)[-1]
        errorf = StringIO()
        traceback.print_exception(exc_type, exc_obj, exc_tb, file=errorf)
        errmsg = ('[{srcfilename}:{lineno}] ({f5filename}#{read_id}) Unhandled '
                  'exception {name}: {msg}\n{exc}'.format(
                      srcfilename=srcfilename, lineno=exc_tb.tb_lineno, f5filename=f5filename,
                      read_id=read_id, name=type(exc).__name__, msg=str(exc),
                      exc=errorf.getvalue()))
        return {'filename': f5filename, 'read_id': read_id, 'status': 'unknown_error',
                'error_message': errmsg}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        pass


class SignalAnalysis:

    def __init__(self, npread, analyzer):
        self.npread = npread
        self.config = analyzer.config
        self.analyzer = proxy(analyzer)

    def set_error(self, error):
        self.npread.set_error(error['status'], error['error_message'])

    def is_stopped(self):
        return self.npread.is_stopped()

    def clear_cache(self):
        self.npread.close()

    def segments_of(self, record):
        """pxg_read_result -> {state name: (first, last)} (signal_analyzer.py:354-362)."""
        names = self.analyzer.ctx.state_names
        return {names[i]: (int(record['seg_first'][i]), int(record['seg_last'][i]))
                for i in range(len(names)) if record['seg_first'][i] >= 0}

    def process(self):
        """Stage order and status/label rules of signal_analyzer.py:230-286."""
        stride = self.config['signal_processing']['rough_signal_stride']
        rec = self.npread.native
        try:
            segments = self.segments_of(rec)
            if 'adapter' not in segments:
                raise SignalAnalysisError('adapter_not_detected')
            if self.config['barcoding']:
                self.push_barcode_signal(None, segments)
            if self.config['measure_polya']:
                self.analyzer.polyaanalyzer(self.npread)
            events = self.load_events()
            if self.config['trim_adapter']:
                self.trim_adapter(events, segments, stride)
            if self.config['filter_unsplit_reads']:
                if self.detect_unsplit_read(events, segments, stride):
                    raise SignalAnalysisError('unsplit_read')
            if self.npread.sequence is not None:
                readlength = len(self.npread.sequence[0]) - self.npread.sequence[2]
                if readlength < self.config['minimum_sequence_length']:
                    raise SignalAnalysisError('sequence_too_short')
        except SignalAnalysisError as exc:
            outname = 'artifact' if exc.args[0] in ('unsplit_read',) else 'fail'
            self.npread.set_status(exc.args[0], stop=True)
            self.npread.set_label(outname)
        else:
            self.npread.set_label('pass')

    def load_events(self):
        if self.config['albacore_onthefly']:
            raise NotImplementedError('on-the-fly albacore basecalling is out of scope')
        bcall = self.npread.load_fast5_events()
        if self.npread.scaling_params is None:
            raise Exception('Signal scaling is not available yet.')
        if not self.config['filter_unsplit_reads']:
            return bcall            # nothing downstream reads the table itself
        return self.event_frame(bcall)

    def event_frame(self, bcall):
        """Base-space columns of the Guppy event table (fast5_file.py:183-208,
        signal_analyzer.py:319-324).  The signal-space columns (mean,
        scaled_mean) live on the GPU: SignalLoader.scan_unsplit_candidates."""
        first, n_blocks, stride = self.npread.guppy_event_geometry()
        moves = np.asarray(bcall['move'], dtype=np.uint8)
        pos = moves.cumsum() - 1
        kmer_size = len(bcall['sequence']) - int(moves.sum()) + 1
        qual = 1 - 10 ** -((np.frombuffer(bcall['qstring'].encode(), 'B') - 33) / 10)
        if kmer_size == 5:          # Guppy old models
            posshift = 2
        elif kmer_size == 1:        # Guppy flip-flop models
            posshift = 0
        else:
            raise Exception('Move table is encoded with an unknown kmer-size.')
        start = np.arange(first, first + stride * n_blocks, stride)
        return {'start': start, 'end': start + np.hstack((np.diff(start), [1])).astype(np.int64),
                'move': moves, 'pos': np.cumsum(moves), 'p_model_state': qual[pos + posshift]}

    def trim_adapter(self, events, segments, elspan):
        # signal_analyzer.py:328-331: returns as soon as a sequence is present,
        # which is always after load_events -> adapter trimming is a no-op in
        # this revision of the reference.
        if self.npread.sequence is not None:
            return

    def detect_unsplit_read(self, events, segments, elspan):
        """Decision rule of signal_analyzer.py:366-443 over the candidate
        in-read adapters the GPU window scan found for this read."""
        try:
            payload_start = (segments['adapter'][1] + 1) * elspan
        except (KeyError, IndexError):
            return False            # must be an adapter-only read
        if self.npread.native_unsplit_count > native.PXG_MAX_UNSPLIT:
            raise Exception('more than {} in-read adapter candidates'.format(native.PXG_MAX_UNSPLIT))
        excessive_adapters = self.npread.native_unsplit
        if not excessive_adapters:
            return False

        config = self.config['unsplit_read_detection']
        adapter_intervals = ([[0, payload_start]] + union_intervals(excessive_adapters)
                             + [[np.inf, np.inf]])
        basequality_cutoff = config['basecount_quality_limit']
        start, pos, pms = events['start'], events['pos'], events['p_model_state']

        def count_high_quality_reads(left, right):
            # events[start.between(left, right)].groupby('pos')['p_model_state'].max() > cutoff
            sel = (start >= left) & (start <= right)
            if not sel.any():
                return 0
            p, q = pos[sel], pms[sel]
            heads = np.nonzero(np.r_[True, p[1:] != p[:-1]])[0]
            return int((np.maximum.reduceat(q, heads) > basequality_cutoff).sum())

        subread_lengths = [count_high_quality_reads(left, right)
                           for (_, left), (right, _) in zip(adapter_intervals[0:],
                                                            adapter_intervals[1:])]
        subread_hq_length_total = sum(subread_lengths[1:])
        return bool(subread_hq_length_total > config['subread_basecount_limit'] or
                    (subread_hq_length_total + 1) / (subread_lengths[0] + 1)
                    > config['subread_baseratio_limit'])

    def detect_segments(self, signal, elspan):
        """Single-read debug path (signal_analyzer.py:346-364) through the GPU hook."""
        scan_limit = self.config['segmentation']['segmentation_scan_limit'] // elspan
        if len(signal) > scan_limit:
            signal = signal[:scan_limit]
        first, last, _, _ = self.analyzer.ctx.viterbi([signal])
        names = self.analyzer.ctx.state_names
        return {names[i]: (int(first[0][i]), int(last[0][i]))
                for i in range(len(names)) if first[0][i] >= 0}

    def push_barcode_signal(self, signal, segments):
        self.analyzer.demuxer.push(self.npread, signal)
