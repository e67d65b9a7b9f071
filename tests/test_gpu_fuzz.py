"""Seeded fuzz of the whole path against the oracle (-m gpu): ragged batches that mix
degenerate lengths (0, 1, 14, 15 samples ... just over every threshold ... 250 000 samples),
structured squiggles, flat lines, full-scale noise and saturated int16 runs, with odd DAQ
calibrations (non-integer offsets, other sampling rates).  Every field of every record, the
poly(A) spike rows and the chimera candidates must equal the oracle's."""
import os

import numpy as np
import pytest

from conftest import assert_spikes_equal

from poreplex_amd import native as N
from poreplex_amd.synth import synth_batch

pytestmark = pytest.mark.gpu

LENGTHS = [0, 1, 14, 15, 16, 100, 3899, 3900, 4500, 8985, 8999, 9000, 9014, 9015, 12000, 29999, 30000,
           30001, 45000, 99990, 100000, 100001, 100015, 130000, 250000]


def make_batch(rng, n):
    pool = synth_batch(24, seed=int(rng.integers(1 << 30)), samples_per_read=int(rng.choice([12000, 40000, 110000])),
                       jitter=0.3, scale_sigma=0.12, shift_sigma=8.0)
    parts, cal = [], []
    for _ in range(n):
        L = int(rng.choice(LENGTHS)) if rng.random() < 0.6 else int(rng.integers(0, 60000))
        kind = rng.integers(0, 6)
        if kind <= 2:                                   # structured squiggle, cut or tiled to L
            i = int(rng.integers(0, 24))
            src = pool['arena'][pool['offsets'][i]:pool['offsets'][i + 1]]
            raw = np.resize(src, L) if L else src[:0]
            c = pool['calib'][i].copy()
        else:
            c = np.zeros(1, dtype=N.CALIB_DTYPE)[0]
            c['range'], c['digitisation'] = rng.uniform(900, 1500), rng.choice([8192.0, 4096.0, 2048.0])
            c['offset'], c['sampling_rate'] = rng.uniform(-40, 40), rng.choice([3012.0, 4000.0, 5000.0])
            if kind == 3:                               # flat line with a little noise
                raw = (rng.integers(300, 900) + rng.normal(0, 3, L)).astype(np.int16)
            elif kind == 4:                             # full-scale noise
                raw = rng.integers(-32768, 32768, L).astype(np.int16)
            else:                                       # saturated runs
                raw = np.repeat(rng.choice([-32768, 32767, 0, 500], max(L // 50, 1)), 50)[:L].astype(np.int16)
                raw = np.resize(raw, L)
        if rng.random() < 0.3:
            c['offset'] = float(c['offset']) + 0.37      # a non-integer DAQ offset
        parts.append(np.ascontiguousarray(raw, dtype=np.int16))
        cal.append(c)
    arena, off = N.pack_reads(parts)
    return arena, off, np.array(cal, dtype=N.CALIB_DTYPE), parts


def test_fuzz_whole_path_vs_oracle(ctx, oracle):
    rng = np.random.default_rng(int(os.environ.get('PXG_FUZZ_SEED', 20260928)))
    mask = N.STAGE_ALL_DEMUX | N.STAGE_POLYA
    seen = np.zeros(len(N.STATUS_NAMES), dtype=np.int64)
    for trial in range(int(os.environ.get('PXG_FUZZ_TRIALS', 40))):
        n = int(rng.choice([1, 3, 17, 64, 130]))
        arena, off, cal, parts = make_batch(rng, n)
        inject = None
        if rng.random() < 0.25:        # caller-supplied scaling (the scaler stage is skipped)
            inject = np.stack([rng.normal(1.0, 0.15, n), rng.normal(0.0, 10.0, n)], axis=1).astype(np.float32)
        want, wsp = oracle.process_batch(arena, off, cal, inject, mask, want_spikes=True)
        ctx.upload(arena, off, cal, inject)
        ctx.run(mask)
        got = ctx.download()
        for f in got.dtype.names:
            assert np.array_equal(got[f], want[f], equal_nan=True), (trial, f, np.nonzero(got[f] != want[f])[0][:5])
        assert_spikes_equal(ctx.download_spikes(), want, wsp, trial)
        seen += np.bincount(got['status'], minlength=len(seen))
        # the chimera scan with Guppy frames of random phase / truncated tables
        first = rng.integers(0, 50, n)
        nb = np.maximum((np.diff(off) - first) // 15 - rng.integers(0, 3, n) * 200, 0)
        iv, cnt, start = ctx.unsplit_scan(first, nb)
        a = 3
        for r in range(n):
            if nb[r] <= 0 or got[r]['status'] != 0 or got[r]['seg_first'][a] < 0:
                assert cnt[r] == 0, (trial, r)
                continue
            _, sc = oracle.guppy_event_means(parts[r], cal[r], int(first[r]), int(nb[r]), got[r]['scale'], got[r]['shift'])
            wiv, wc = oracle.unsplit_scan(sc, int(first[r]), (int(got[r]['seg_last'][a]) + 1) * 15,
                                          float(cal[r]['sampling_rate']))
            assert cnt[r] == wc and iv[start[r]:start[r + 1]].tolist() == wiv.tolist(), (trial, r)
    # the fuzz reached the failure statuses as well as the happy path
    assert seen[N.STATUS_CODE['okay']] > 20 and seen[N.STATUS_CODE['scaler_signal_too_short']] > 20
    assert seen[N.STATUS_CODE['scaling_qc_fail']] + seen[N.STATUS_CODE['adapter_not_detected']] > 5


def random_left_to_right_model(rng):
    """2..8 states in chain order, random spans / mixtures / start states, names shuffled (the
    in-edge order of a tie is the name order), and CLONED states -- same emission, same in- and
    out-edges -- so that exact ties between candidates happen long after the first steps."""
    S = int(rng.integers(2, 9))
    pool = ['adapter', 'polya-tail', 'zeta', 'alpha', 'mid', 'beta', 'omega', 'kappa']
    names = [str(x) for x in rng.permutation(pool)[:S]]
    max_span = int(rng.choice([1, 2, 3, 5, 7]))
    mus = rng.uniform(60, 130, S)
    states = []
    for i, name in enumerate(names):
        n_mix = int(rng.choice([1, 1, 2, 3]))
        em = [[float(mus[i] + rng.normal(0, 6)), float(rng.uniform(2, 9)), float(rng.uniform(0.2, 1.0))]
              for _ in range(n_mix)]
        if n_mix == 1:
            em = [em[0][:2]]
        targets = [i] + [j for j in range(i + 1, min(S, i + max_span + 1)) if rng.random() < 0.7]
        if i + 1 < S and i + 1 not in targets:
            targets.append(i + 1)
        if rng.random() < 0.4:           # equal probabilities: candidates tie as often as they can
            p = np.full(len(targets), 1.0 / len(targets))
        else:
            p = rng.dirichlet(np.ones(len(targets)) * 2)
        st = {'name': name, 'emission': em, 'transition': [[names[j], float(q)] for j, q in zip(targets, p)]}
        states.append(st)
    n_start = int(rng.integers(1, min(S, 3) + 1))
    sp = rng.dirichlet(np.ones(n_start)) if rng.random() < 0.6 else np.full(n_start, 1.0 / n_start)
    for i in range(n_start):
        states[i]['start_prob'] = float(sp[i])
    if S >= 4 and rng.random() < 0.7:    # state 2 becomes a clone of state 1 (both fed by state 0, both feed state 3)
        states[2]['emission'] = [list(e) for e in states[1]['emission']]
        states[0]['transition'] = [[names[0], 0.5], [names[1], 0.25], [names[2], 0.25]]
        states[1]['transition'] = [[names[1], 0.75], [names[3], 0.25]]
        states[2]['transition'] = [[names[2], 0.75], [names[3], 0.25]]
        if n_start > 1:
            for st in states[1:]:
                st.pop('start_prob', None)
            states[0]['start_prob'] = 1.0
    return states, mus


def test_fuzz_random_left_to_right_hmms_vs_oracle(config):
    """K3's generic template (2- and 4-bit back-pointer fields, tie replay) on random models,
    through the pooled-signal hook: paths identical to the oracle's, log-probabilities to 1e-9.
    (Negative control, run by hand: with the exact replay of tied chunks compiled out the first
    model already fails, so the ties are there; 900 models over three seeds pass.)"""
    import copy
    from oracle.pxo import Oracle
    rng = np.random.default_rng(int(os.environ.get('PXG_FUZZ_SEED', 7)))
    for trial in range(int(os.environ.get('PXG_FUZZ_MODELS', 16))):
        cfg = copy.deepcopy(config)
        cfg['segmentation_model'], mus = random_left_to_right_model(rng)
        orc = Oracle(cfg)
        c = N.NativeContext(cfg, device_id=0)
        try:
            sigs = []
            for n in (1, 2, 15, 16, 17, 33, 400, int(rng.integers(500, 3000)), 7000):
                k = max(1, n // int(rng.integers(20, 200)))         # piece-wise levels drawn from the states' means
                levels = np.repeat(rng.choice(mus, k + 1), n // k + 1)[:n]
                sigs.append((levels + rng.normal(0, 3, n)).astype(np.float32))
            first, last, paths, logp = c.viterbi(sigs, want_path=True)
            for k, sg in enumerate(sigs):
                olp, opath = orc.viterbi(sg)
                assert np.array_equal(paths[k], opath), (trial, k, cfg['segmentation_model'])
                assert abs(logp[k] - olp) <= 1e-9 * max(1.0, abs(olp)), (trial, k)
                ofirst, olast = orc.segments(opath)
                assert np.array_equal(first[k], ofirst) and np.array_equal(last[k], olast), (trial, k)
        finally:
            c.close()


def test_fuzz_stage_hooks_vs_oracle(ctx, oracle):
    """The per-stage hooks on adversarial inputs: barcode windows full of duplicated values, signed
    zeros and constant runs (the medians are radix selects over order-preserving keys), windows of
    every length around the 260 / 300 / 3000 gates; event detection on constant, stepped, spiky
    and tiny windows; DAQ -> pA / pooling on full-scale int16 with odd calibrations."""
    rng = np.random.default_rng(int(os.environ.get('PXG_FUZZ_SEED', 3)))
    # ---- a9-a11 ------------------------------------------------------------------
    sigs = []
    for n in [0, 1, 259, 260, 261, 299, 300, 301, 2999, 3000, 3001] + rng.integers(200, 3400, 60).tolist():
        kind = rng.integers(0, 5)
        if kind == 0:
            x = rng.normal(90, 12, n)
        elif kind == 1:
            x = rng.choice([80.0, 80.0, 95.5, 101.25], n)                   # heavy ties
        elif kind == 2:
            x = np.full(n, 77.0)                                            # MAD = 0 -> the 0.01 floor
        elif kind == 3:
            x = rng.choice([0.0, -0.0, 1e-30, -1e-30, 3.5], n)              # signed zeros, denormal-ish
        else:
            x = np.round(rng.normal(90, 4, n))                              # integers: many equal pairs
        sigs.append(x.astype(np.float32))
    out, pushed = ctx.barcode_window(sigs)
    for k, sg in enumerate(sigs):
        want, wp = oracle.barcode_window(sg)
        assert bool(pushed[k]) == bool(wp), (k, len(sg))
        if wp:
            assert np.array_equal(out[k].view(np.uint32), want.view(np.uint32)), (k, len(sg))
    # ---- a15 -----------------------------------------------------------------------
    wins = []
    for n in [1, 2, 3, 6, 7, 13, 14, 19, 20, 21, 39, 40, 41, 63, 64, 65] + rng.integers(30, 6000, 50).tolist():
        kind = rng.integers(0, 5)
        if kind == 0:
            x = np.full(n, 100.0)
        elif kind == 1:
            x = np.repeat(rng.normal(95, 15, n // 3 + 1), 3)[:n]            # steps shorter than both windows
        elif kind == 2:
            x = rng.normal(95, 1, n); x[rng.integers(0, n, max(n // 50, 1))] += 400   # spikes
        elif kind == 3:
            x = np.repeat(rng.normal(95, 15, n // 40 + 1), 40)[:n] + rng.normal(0, 0.05, n)
        else:
            x = rng.normal(0, 1e-3, n)                                       # variance at the FLT_MIN clamp
        wins.append(x.astype(np.float32))
    evs, cnt = ctx.detect_events(wins, max_events=2500)
    for k, w in enumerate(wins):
        want = oracle.detect_events(w)
        assert cnt[k] == len(want), (k, len(w))
        for f in ('start', 'length', 'mean', 'stdv'):
            assert np.array_equal(evs[k][f], want[f], equal_nan=True), (k, f, len(w))
    # ---- a1, a2, a5 ----------------------------------------------------------------
    parts, cal = [], np.zeros(24, dtype=N.CALIB_DTYPE)
    for i in range(24):
        n = int(rng.choice([0, 14, 15, 8999, 9000, 30000, 30014, 31000, 47000]))
        parts.append(rng.integers(-32768, 32768, n).astype(np.int16))
        cal[i] = (rng.uniform(100, 3000), rng.choice([2048.0, 8192.0, 65536.0]), rng.uniform(-500, 500),
                  rng.choice([3012.0, 4000.0]))
    arena, off = N.pack_reads(parts)
    head, status = ctx.head_pool(arena, off, cal)
    ss = np.stack([rng.normal(1, 0.2, 24), rng.normal(0, 20, 24)], axis=1).astype(np.float32)
    pooled, poff = ctx.pool_scale(arena, off, cal, ss)
    for i, raw in enumerate(parts):
        wh, wst = oracle.head_pool(raw, cal[i])
        assert status[i] == wst, i
        if wst == 0:
            assert np.array_equal(head[i].view(np.uint32), wh.view(np.uint32)), i
        assert np.array_equal(pooled[poff[i]:poff[i + 1]].view(np.uint32),
                              oracle.pool_scale(raw, cal[i], ss[i, 0], ss[i, 1]).view(np.uint32)), i
        if len(raw):
            assert np.array_equal(ctx.raw_to_pa(raw, cal[i]).view(np.uint32),
                                  oracle.raw_to_pa(raw, cal[i]).view(np.uint32)), i


def test_fuzz_lstm_hooks_on_extreme_inputs(ctx, oracle):
    """The two networks on inputs that drive the spline tables to and past their ends (the sigmoid
    table covers [-32, 32), tanh's half of that): huge and tiny values, exact zeros, constant rows,
    alternating signs, the -1000 padding everywhere.  Bit-identical to the oracle."""
    rng = np.random.default_rng(int(os.environ.get('PXG_FUZZ_SEED', 21)))
    width = oracle.cfg.scaler_length // oracle.cfg.stride

    def rows(n, T):
        out = np.zeros((n, T), dtype=np.float32)
        for i in range(n):
            kind = i % 8
            if kind == 0:
                out[i] = rng.normal(90, 15, T)
            elif kind == 1:
                out[i] = rng.choice([-1e6, 1e6, 0.0, 3e4, -3e4], T)
            elif kind == 2:
                out[i] = 0.0
            elif kind == 3:
                out[i] = np.where(np.arange(T) % 2 == 0, 500.0, -500.0)
            elif kind == 4:
                out[i] = rng.normal(0, 1e-20, T)
            elif kind == 5:
                out[i] = -1000.0
            elif kind == 6:
                out[i, T // 2:] = rng.normal(100, 200, T - T // 2)      # left-padded like a short head
            else:
                out[i] = np.linspace(-40, 40, T)
        return out
    heads = rows(19, width)
    got = ctx.scaler_lstm(heads)
    want = np.stack([oracle.scaler_forward(h) for h in heads])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    wins = rows(35, oracle.cfg.signal_trim_length)
    got = ctx.demux_lstm(wins)
    want = np.stack([oracle.demux_forward(w) for w in wins])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    assert np.isfinite(got).all()
