"""Poly(A) tail measurement facade (reference: poreplex/polya.py).

Event detection (a15), the interval DP (a16) and the retry/recalibration logic
(a14, a17) are GPU stages; this class converts the per-read record into the
dict the reference stores with ``npread.set_polya_tail`` (polya.py:116-121).
"""
__all__ = ['PolyASignalAnalyzer']


class PolyASignalAnalyzer:

    def __init__(self, config, ctx):
        self.ctx = ctx
        vars(self).update(config)        # the polya_dwell options, by their config names
        centre, spread = config['polya_mean_dist']
        halfwidth = spread * config['polya_mean_z_cutoff']
        self.polya_mean_cutoff = (centre - halfwidth, centre + halfwidth)

    def __call__(self, npread, rough_range=None, stride=None):
        rec = npread.native
        if rec is None or not rec['polya_called']:
            return
        n_spikes = int(rec['polya_n_spikes'])
        rows = npread.native_spikes
        if rows is None:
            n_spikes = 0
        elif n_spikes > len(rows):
            # the GPU keeps PXG_MAX_SPIKES rows per read; the reference lists every spike,
            # so a longer list cannot be reported faithfully: fail this read, loudly
            raise Exception('poly(A) tail with more than {} spike events'.format(len(rows)))
        npread.set_polya_tail({
            'begin': int(rec['polya_begin']),
            'end': int(rec['polya_end']),
            'dwell_time': int(rec['polya_dwell_samples']) / npread.sampling_rate,
            'spikes': [tuple(row) for row in rows[:n_spikes].astype(float).tolist()]
                      if n_spikes else [],
        })
