"""Poly(A) tail measurement facade (reference: poreplex/polya.py).

Event detection (a15), the interval DP (a16) and the retry/recalibration logic
(a14, a17) are GPU stages; this class converts the per-read record into the
dict the reference stores with ``npread.set_polya_tail`` (polya.py:116-121).
"""
import numpy as np

__all__ = ['PolyASignalAnalyzer']


class PolyASignalAnalyzer:

    def __init__(self, config, ctx):
        self.ctx = ctx
        vars(self).update(config)        # the polya_dwell options, by their config names
        centre, spread = config['polya_mean_dist']
        halfwidth = spread * config['polya_mean_z_cutoff']
        self.polya_mean_cutoff = (centre - halfwidth, centre + halfwidth)

    def assign(self, table, rows, records):
        """set_polya_tail for many reads at once, as columns: begin / end / dwell time / spike
        count of every called tail go into the table, and the dict of polya.py:116-121 is built
        from them (and the spike rows: all of them, the GPU keeps every spike) when somebody asks
        for it (ReadTable.polya_of).  Returns the positions that need the per-read path: none."""
        called = np.nonzero(records['polya_called'])[0]
        at = rows[called]
        table.polya_begin[at] = records['polya_begin'][called]
        table.polya_end[at] = records['polya_end'][called]
        table.polya_dwell_time[at] = records['polya_dwell_samples'][called] / table.sampling_rate[at]
        table.polya_spike_count[at] = records['polya_n_spikes'][called] if table.spikes is not None else 0
        table.polya_lazy[at] = True
        return []

    def __call__(self, npread, rough_range=None, stride=None):
        rec = npread.native
        if rec is None or not rec['polya_called']:
            return
        rows = npread.native_spikes
        n_spikes = 0 if rows is None else int(rec['polya_n_spikes'])
        npread.set_polya_tail({
            'begin': int(rec['polya_begin']),
            'end': int(rec['polya_end']),
            'dwell_time': int(rec['polya_dwell_samples']) / npread.sampling_rate,
            'spikes': [tuple(row) for row in rows[:n_spikes].astype(float).tolist()]
                      if n_spikes else [],
        })
