"""Barcode demultiplexer facade (reference: poreplex/barcoding.py).

The numerics (robust z-score window a10/a11, the recurrent classifier a12,
threshold and calibrated phred a13) run on the GPU inside the batch call; this
class keeps the reference's operator surface and turns the per-read records
into ``npread.set_barcode(...)`` calls (barcoding.py:108-118).
"""
import numpy as np

__all__ = ['BarcodeDemultiplexer']


class BarcodeDemultiplexer:

    PAD_FILLER = -1000.

    def __init__(self, config, qualitythreshold, ctx):
        self.config, self.ctx = config, ctx
        self.calibration_table = [float(v) for v in ctx.ncfg.calibration]
        if len(self.calibration_table) - 1 < qualitythreshold:     # barcoding.py:41-44
            raise ValueError('The current demultiplexer does not support calibrated score '
                             'of {}. Consider lowering --barcoding-quality-filter value.'
                             .format(qualitythreshold))
        self.score_threshold = self.calibration_table[qualitythreshold]
        self.signal_assoc_read = []

    def clear(self):
        del self.signal_assoc_read[:]

    def lookup_calibrated_phred_score(self, score):
        from bisect import bisect_right
        if score <= 0.:
            return 0
        return bisect_right(self.calibration_table, float(score))

    def normalize_signal(self, sig):
        """Single-window debug path through the GPU hook (barcoding.py:77-81);
        only defined for windows that pass the length gate."""
        out, pushed = self.ctx.barcode_window([np.asarray(sig, dtype=np.float32)])
        if not pushed[0]:
            raise ValueError('window outside minimum/maximum_dna_length')
        n = min(len(sig), self.config['signal_trim_length'])
        return out[0][-n:]

    def push(self, npread, signal=None):
        """Queue a read whose adapter window passed the gate (decided on the
        GPU: record field bc_pushed, barcoding.py:84-88)."""
        if npread.native is not None and npread.native['bc_pushed']:
            self.signal_assoc_read.append(npread)

    def predict(self):
        for npread in self.signal_assoc_read:
            rec = npread.native
            bcid, score = int(rec['bc_label']), np.float32(rec['bc_score'])
            effective = bcid if rec['bc_called'] else None
            npread.set_barcode(effective, bcid, int(rec['bc_phred']))
            npread.barcode_probs = np.array(rec['probs'][:self.ctx.cfg.demux_dense.out_dim])
            npread.barcode_raw_score = score
