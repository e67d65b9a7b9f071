"""Handle onto the GPU-resident segmentation HMM.

Stands in for the pomegranate model object the reference builds in
worker_persistence.py:95-121 and calls at signal_analyzer.py:352,389:
``viterbi(signal)`` keeps pomegranate's return shape -- (logp, [(idx, state)...])
with the silent start state first -- as a single-read debug path; the batch
path never goes through Python per read.
"""
import numpy as np

__all__ = ['SegmentationModel']


class _State:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return '<State {}>'.format(self.name)


class SegmentationModel:

    def __init__(self, ctx, which):
        self.ctx, self.which = ctx, which
        names = ctx.ncfg.unsplit_state_names if which else ctx.state_names
        self.states = [_State(n) for n in names]
        self.start = _State('model-start')

    def viterbi(self, signal):
        signal = np.ascontiguousarray(signal, dtype=np.float32)
        _, _, paths, logp = self.ctx.viterbi([signal], which_model=self.which, want_path=True)
        calls = [(len(self.states), self.start)]
        calls += [(int(s), self.states[int(s)]) for s in paths[0]]
        return float(logp[0]), calls
