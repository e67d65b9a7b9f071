"""Poly(A) tail measurement facade (reference: poreplex/polya.py).

Event detection (a15), the interval DP (a16) and the retry/recalibration logic
(a14, a17) are GPU stages; this class converts the per-read record into the
dict the reference stores with ``npread.set_polya_tail`` (polya.py:116-121).
"""
__all__ = ['PolyASignalAnalyzer']


class PolyASignalAnalyzer:

    CONFIG_SLOTS = [
        'refinement_expansion', 'event_detection', 'polya_stdv_max', 'polya_stdv_range',
        'spike_tolerance', 'spike_weight', 'openend_expansion', 'recalibrate_shifted_signal',
        'polya_mean_dist', 'polya_mean_z_cutoff', 'polya_mean_trigger_recalibration',
        'maximum_openend_extension', 'median_pre_filter',
    ]

    def __init__(self, config, ctx):
        self.ctx = ctx
        for name in self.CONFIG_SLOTS:
            setattr(self, name, config[name])
        mean_loc, mean_scale = config['polya_mean_dist']
        self.polya_mean_cutoff = (mean_loc - mean_scale * config['polya_mean_z_cutoff'],
                                  mean_loc + mean_scale * config['polya_mean_z_cutoff'])

    def __call__(self, npread, rough_range=None, stride=None):
        rec = npread.native
        if rec is None or not rec['polya_called']:
            return
        spikes = []
        if npread.native_spikes is not None:
            for k in range(int(rec['polya_n_spikes'])):
                row = npread.native_spikes[k]
                spikes.append((float(row[0]), float(row[1]), float(row[2]), float(row[3])))
        npread.set_polya_tail({
            'begin': int(rec['polya_begin']),
            'end': int(rec['polya_end']),
            'dwell_time': int(rec['polya_dwell_samples']) / npread.sampling_rate,
            'spikes': spikes,
        })
