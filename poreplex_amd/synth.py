"""Seeded synthetic direct-RNA reads.

No FAST5 sample exists anywhere near the reference, so tests and bench.py use
piece-wise stationary squiggles drawn from the segmentation HMM's own
emissions (presets/rna-r941.cfg segmentation_model): pre-leader, leader-low,
leader-high, adapter (optionally ending in a barcode prototype), poly(A),
transcript.  Levels are held for geometric dwells so the signal looks
event-like, then pushed back through a per-read (scale, shift) and the DAQ
quantiser so the int16 samples exercise a1-a5 exactly like a real read.
"""
import os

import numpy as np

from .native import CALIB_DTYPE

# emission parameters, rna-r941.cfg:61-101 (mu, sigma[, weight])
_EMIT = {
    'pre-leader': [(71.50333676145819, 3.661488091442809, 1.0)],
    'leader-low': [(102.06822592375592, 3.906628405239109, 1.0)],
    'leader-high': [(112.02391752604612, 4.798575215964485, 1.0)],
    'adapter': [(80.49361662338703, 7.40645452440055, 0.8480750651871221),
                (65.30177304787502, 3.305209545883456, 0.15192493481287794)],
    'polya-tail': [(108.95443922911944, 2.549021552095772, 1.0)],
    'transcript': [(81.90432308474264, 7.760903146911895, 0.49045594516210067),
                   (109.72965574809056, 12.733734156887294, 0.5095440548378992)],
}
_PIECES = ('pre-leader', 'leader-low', 'leader-high', 'adapter', 'polya-tail', 'transcript')
_PIECE_LEN = {'pre-leader': (200, 600), 'leader-low': (300, 600), 'leader-high': (200, 500),
              'adapter': (4000, 9000), 'polya-tail': (600, 3000)}
PROTOTYPE_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'presets',
                              'MIN106-RNA001', 'synthetic-barcode-prototypes.npy')


def load_prototypes():
    """5 x 300 float32 z-score windows (row 0 = decoy class, unused) or None.  Made by
    tools/make_prototypes.py: windows the shipped demux net classifies with p > 0.995
    (checked with the oracle), so synthetic reads carry callable barcodes."""
    if os.path.isfile(PROTOTYPE_FILE):
        return np.load(PROTOTYPE_FILE).astype(np.float32)
    return None


def _draw_levels(rng, state_idx):
    """One emission draw per event from the state's (mixture of) normals."""
    out = np.empty(state_idx.shape, dtype=np.float32)
    for si, name in enumerate(_PIECES):
        m = state_idx == si
        n = int(m.sum())
        if not n:
            continue
        comps = _EMIT[name]
        if len(comps) == 1:
            mu, sd, _ = comps[0]
            out[m] = rng.normal(mu, sd, n)
        else:
            w = np.array([c[2] for c in comps])
            pick = rng.choice(len(comps), size=n, p=w / w.sum())
            mu = np.array([c[0] for c in comps])[pick]
            sd = np.array([c[1] for c in comps])[pick]
            out[m] = rng.normal(mu, sd)
    return out


def synth_batch(n_reads, seed=922, samples_per_read=60000, jitter=0.1, barcodes=None,
                prototypes='auto', mean_dwell=9.0, sample_noise=1.5, with_polya=True,
                short_fraction=0.0, scale_sigma=0.05, shift_mu=-5.0, shift_sigma=3.0,
                fixed_calib=False, adversarial=None, length_dist=None):
    """Generate a ragged batch.

    adversarial (dict, optional; drawn from a SEPARATE generator so every other batch keeps its bits):
    reads built to sit ON the decisions instead of inside them (tests/test_decision_flips.py,
    tools/decision_flips_gpu.py) --
      'blend':  per read and state, with probability 1/2 the state's levels are shifted this fraction of
                the way to the NEXT state's mean (0.5 = midway between neighbouring states);
      'drift':  a linear drift of up to +-drift pA over the read;
      'adapter_gate': with this probability (True = always) a read's adapter piece is made 260 or 3 000
                pooled samples long +- a few (the gates of BarcodeDemultiplexer.push, barcoding.py:87-88)
                instead of 4 000 - 9 000 raw samples.

    length_dist='lognormal': read lengths as a sequencing run produces them instead of samples_per_read
    +- jitter (the reference takes whatever the flow cell wrote, pipeline.py:303-337): 95 % log-normal around
    a median of 40 000 samples with 15 % of the reads below 30 000 (so the scaler's zero padding and the
    9 000-sample gate are in play), 5 % from a heavy tail up to 1 000 000 samples; clipped to [6 000, 1 000 000].

    Returns dict(arena int16, offsets int64[n+1], calib CALIB_DTYPE[n],
    scale_shift float32[n,2] (the TRUE per-read scaling: pA_model =
    scale*pA_raw + shift), barcode int8[n] (-1 decoy / 0..3), truth int32[n,6,2]
    piece boundaries in raw samples).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    if isinstance(prototypes, str) and prototypes == 'auto':
        prototypes = load_prototypes()
    if barcodes is None:
        barcodes = rng.integers(-1, 4, size=n_reads).astype(np.int8)
    else:
        barcodes = np.asarray(barcodes, dtype=np.int8)

    lens = np.maximum(
        (samples_per_read * (1.0 + jitter * rng.uniform(-1, 1, n_reads))).astype(np.int64),
        16)
    if length_dist == 'lognormal':
        lrng = np.random.Generator(np.random.PCG64(seed + 104729))
        body = 40000.0 * np.exp(0.2776 * lrng.standard_normal(n_reads))
        tail = 40000.0 * np.exp(0.5 + 1.1 * np.abs(lrng.standard_normal(n_reads)))
        lens = np.clip(np.where(lrng.random(n_reads) < 0.05, tail, body), 6000, 1000000).astype(np.int64)
    elif length_dist is not None:
        raise ValueError('unknown length_dist')
    n_short = int(round(short_fraction * n_reads))
    if n_short:
        lens[rng.choice(n_reads, n_short, replace=False)] = rng.integers(2000, 8000, n_short)
    offsets = np.zeros(n_reads + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(lens)
    arena = np.empty(int(offsets[-1]), dtype=np.int16)

    # piece boundaries (raw samples); the adapter end is aligned to the stride
    bounds = np.zeros((n_reads, 7), dtype=np.int64)
    cur = np.zeros(n_reads, dtype=np.int64)
    for pi, name in enumerate(_PIECES[:5]):
        lo, hi = _PIECE_LEN[name]
        plen = rng.integers(lo, hi + 1, n_reads)
        if name == 'polya-tail' and not with_polya:
            plen[:] = 0
        cur = cur + plen
        if name == 'adapter':
            cur -= cur % 15
        bounds[:, pi + 1] = cur
    adv = adversarial or {}
    arng = np.random.Generator(np.random.PCG64(seed + 7919)) if adv else None
    if adv.get('adapter_gate'):
        gate = np.where(arng.random(n_reads) < 0.7, 260, 3000) + arng.integers(-3, 4, n_reads)
        a_len = gate * 15 + arng.integers(-14, 15, n_reads)
        p_gate = 1.0 if adv['adapter_gate'] is True else float(adv['adapter_gate'])
        shift_by = np.where(arng.random(n_reads) < p_gate, (bounds[:, 3] + a_len) - bounds[:, 4], 0)
        bounds[:, 4:6] += shift_by[:, None]
    bounds[:, 6] = np.maximum(lens, bounds[:, 5])
    bounds = np.minimum(bounds, lens[:, None])
    if adv.get('blend'):
        mu = np.array([sum(c[0] * c[2] for c in _EMIT[name]) / sum(c[2] for c in _EMIT[name]) for name in _PIECES])
        toward = np.append(mu[1:], mu[-1]) - mu                       # to the next state's mean
        blend_shift = (arng.random((n_reads, 6)) < 0.5) * (float(adv['blend']) * toward)[None, :]
    drift_total = arng.uniform(-1, 1, n_reads) * float(adv['drift']) if adv.get('drift') else None

    scale = rng.normal(0.955, scale_sigma, n_reads).astype(np.float32)
    shift = rng.normal(shift_mu, shift_sigma, n_reads).astype(np.float32)
    calib = np.zeros(n_reads, dtype=CALIB_DTYPE)
    calib['range'] = rng.uniform(1150, 1250, n_reads)
    calib['digitisation'] = 8192.0
    calib['offset'] = np.floor(rng.uniform(-5, 25, n_reads))
    calib['sampling_rate'] = 3012.0
    if fixed_calib:      # same DAQ settings for every read (reads can be concatenated)
        calib['range'], calib['offset'] = 1200.0, 10.0

    chunk = 512
    cuts = list(range(0, n_reads, chunk)) + [n_reads]
    order = np.arange(n_reads)
    if length_dist is not None:
        # reads of similar length together (the event arrays of a chunk are dense, sized by its longest
        # read), chunks of bounded reads x longest read
        order = np.argsort(lens, kind='stable')
        cuts, c0 = [0], 0
        while c0 < n_reads:
            c1 = c0
            while c1 < n_reads and c1 - c0 < chunk and (c1 - c0 + 1) * int(lens[order[c1]]) <= chunk * 80000:
                c1 += 1
            c1 = max(c1, c0 + 1)
            cuts.append(c1)
            c0 = c1
    for c0, c1 in zip(cuts[:-1], cuts[1:]):
        B = c1 - c0
        sel = order[c0:c1]
        L = lens[sel]
        Lmax = int(L.max())
        E = int(Lmax / mean_dwell * 1.25) + 64
        dwell = rng.geometric(1.0 / mean_dwell, size=(B, E)).astype(np.int64)
        starts = np.cumsum(dwell, axis=1) - dwell
        # make sure every read is fully covered
        short = starts[:, -1] + dwell[:, -1] < L
        dwell[short, -1] += (L - (starts[:, -1] + dwell[:, -1]))[short]
        ends = np.minimum(starts + dwell, L[:, None])
        cnt = np.maximum(ends - np.minimum(starts, L[:, None]), 0)
        state = (starts[:, :, None] >= bounds[sel][:, None, 1:6]).sum(axis=2)
        level = _draw_levels(rng, state)
        if adv.get('blend'):
            level = level + np.take_along_axis(blend_shift[sel], state, axis=1).astype(np.float32)
        flat = np.repeat(level.ravel(), cnt.ravel())
        flat += rng.standard_normal(flat.shape[0], dtype=np.float32) * np.float32(sample_noise)
        roff = np.zeros(B + 1, dtype=np.int64)
        roff[1:] = np.cumsum(L)
        if drift_total is not None:
            pos = np.arange(flat.shape[0], dtype=np.float64) - np.repeat(roff[:-1], L)
            flat += (np.repeat(drift_total[sel] / np.maximum(L, 1), L) * pos).astype(np.float32)
        # barcode prototype: last 300 pooled samples of the adapter piece
        if prototypes is not None:
            for b in range(B):
                bc = int(barcodes[sel[b]])
                if bc < 0:
                    continue
                a_end = int(bounds[sel[b], 4])
                a_beg = int(bounds[sel[b], 3])
                w0 = a_end - 300 * 15
                if w0 < a_beg or a_end > L[b]:
                    continue
                proto = 80.49361662338703 + 7.40645452440055 * prototypes[bc + 1]
                seg = np.repeat(proto.astype(np.float32), 15)
                seg = seg + rng.standard_normal(seg.shape[0], dtype=np.float32) * \
                    np.float32(sample_noise)
                flat[roff[b] + w0:roff[b] + a_end] = seg
        # inverse scaling and DAQ quantisation
        rs = np.repeat(scale[sel], L)
        rh = np.repeat(shift[sel], L)
        pa_raw = (flat - rh) / rs
        k = np.repeat(calib['digitisation'][sel] / calib['range'][sel], L)
        off = np.repeat(calib['offset'][sel], L)
        q = np.clip(np.rint(pa_raw * k - off), -32768, 32767).astype(np.int16)
        if length_dist is None:
            arena[offsets[c0]:offsets[c1]] = q
        else:
            for b in range(B):
                arena[offsets[sel[b]]:offsets[sel[b] + 1]] = q[roff[b]:roff[b + 1]]

    truth = np.stack([bounds[:, :6], bounds[:, 1:7]], axis=2).astype(np.int32)
    return {
        'arena': arena, 'offsets': offsets, 'calib': calib,
        'scale_shift': np.stack([scale, shift], axis=1).astype(np.float32),
        'barcode': barcodes, 'truth': truth,
    }


def synth_basecalls(batch, seed=0, block_stride=15):
    """Plausible Guppy basecall summaries for a synthetic batch (one per read): a sequence of
    one base per ~12 blocks, a Move table with one block per 15 samples whose moves add up to
    len(sequence) - 4 (5-mer frames), first_sample_template 0."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for n_raw in np.diff(batch['offsets']).tolist():
        n_blocks = n_raw // block_stride
        n_bases = max(n_blocks // 12, 16)
        move = np.zeros(n_blocks, dtype=np.uint8)
        move[rng.choice(n_blocks, size=min(n_bases - 4, n_blocks), replace=False)] = 1
        n_bases = int(move.sum()) + 4
        seq = ''.join('ACGU'[i] for i in rng.integers(0, 4, n_bases))
        qual = ''.join(chr(33 + q) for q in rng.integers(5, 25, n_bases))
        out.append({'sequence': seq, 'qstring': qual, 'block_stride': block_stride,
                    'sequence_length': n_bases, 'mean_qscore': float(np.float32(rng.uniform(7, 13))),
                    'num_events': n_blocks, 'first_sample_template': 0, 'table': 'move',
                    'move': move})
    return out
