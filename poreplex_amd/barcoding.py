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
        # NativeConfig already refused a quality filter beyond the calibration table
        # (barcoding.py:41-45), with the reference's message
        self.config, self.ctx = config, ctx
        self.calibration_table = [float(v) for v in ctx.ncfg.calibration]
        self.score_threshold = self.calibration_table[qualitythreshold]
        self.signal_assoc_read = []

    def clear(self):
        self.signal_assoc_read = []

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

    def assign(self, table, rows, records):
        """Barcode columns of many reads at once (what push + predict do per read,
        barcoding.py:83-118): only reads whose adapter window passed the length gate
        (decided on the GPU, record field bc_pushed) get a guess and a calibrated score."""
        gate = records['bc_pushed'] != 0
        rows, rec = rows[gate], records[gate]
        called = rec['bc_called'] != 0
        table.has_barcode[rows] = called
        table.barcode[rows] = np.where(called, rec['bc_label'], -1)
        table.barcode_guess[rows] = rec['bc_label']
        table.barcode_phred[rows] = rec['bc_phred']

    def push(self, npread, signal=None):
        """Single-read form of the reference surface: queue the read for predict()."""
        if npread.native is not None and npread.native['bc_pushed']:
            self.signal_assoc_read.append(npread)

    def predict(self):
        for npread in self.signal_assoc_read:
            rec = npread.native
            self.assign(npread.table, np.array([npread.row]), np.array([rec], dtype=rec.dtype))
        self.clear()
