"""Independent cross-checks of the oracle rows that the reference cannot pin
(TensorFlow / pomegranate are not installable here): float64 NumPy LSTM,
torch-CPU nn.LSTM with re-packed Keras weights, brute-force Viterbi."""
import itertools

import numpy as np
import pytest

from poreplex_amd.config import load_model_arrays


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_f64(W, U, b, x, reverse=False):
    """Keras LSTM equations in float64; x [T, I] -> seq [T, H]."""
    W, U, b = W.astype(np.float64), U.astype(np.float64), b.astype(np.float64)
    H = U.shape[0]
    h, c = np.zeros(H), np.zeros(H)
    seq = np.zeros((len(x), H))
    order = range(len(x) - 1, -1, -1) if reverse else range(len(x))
    for t in order:
        z = x[t] @ W + h @ U + b
        i, f, g, o = _sig(z[:H]), _sig(z[H:2 * H]), np.tanh(z[2 * H:3 * H]), _sig(z[3 * H:])
        c = f * c + i * g
        h = o * np.tanh(c)
        seq[t] = h
    return seq


def scaler_f64(head):
    m = load_model_arrays('MIN106-RNA001/scaler-r3.npz')
    s1 = lstm_f64(m['lstm1_kernel'], m['lstm1_recurrent'], m['lstm1_bias'],
                  head.astype(np.float64)[:, None])
    s2 = lstm_f64(m['lstm2_kernel'], m['lstm2_recurrent'], m['lstm2_bias'], s1)
    return s2[-1] @ m['dense_kernel'].astype(np.float64) + m['dense_bias']


def demux_f64(win):
    m = load_model_arrays('MIN106-RNA001/demux-tetra-r4.npz')
    x = win.astype(np.float64)[:, None]
    f = lstm_f64(m['fwd_kernel'], m['fwd_recurrent'], m['fwd_bias'], x)
    b = lstm_f64(m['bwd_kernel'], m['bwd_recurrent'], m['bwd_bias'], x, reverse=True)
    t = lstm_f64(m['top_kernel'], m['top_recurrent'], m['top_bias'], np.concatenate([f, b], 1))
    z = t[-1] @ m['dense_kernel'].astype(np.float64) + m['dense_bias']
    e = np.exp(z - z.max())
    return e / e.sum()


def test_transcendental_kit_accuracy(oracle):
    rng = np.random.default_rng(0)
    x = np.concatenate([np.linspace(-40, 40, 8001), rng.normal(0, 3, 20000)]).astype(np.float32)
    assert np.abs(oracle.sigmoid(x) - _sig(x.astype(np.float64))).max() < 1.0e-7
    assert np.abs(oracle.tanh(x) - np.tanh(x.astype(np.float64))).max() < 2.0e-7
    xe = np.linspace(-80, 80, 4001).astype(np.float32)
    rel = np.abs(oracle.expf(xe) / np.exp(xe.astype(np.float64)) - 1)
    assert rel.max() < 2.0 ** -23
    # saturation: -1000 pad steps must not produce NaN (SURVEY K5 note); the
    # spline saturates at sigmoid(+-32)
    lo, hi, mid = oracle.sigmoid(np.float32([-1e4, 1e4, 0.0]))
    assert 0.0 < lo < 2e-14 and hi == 1.0 and mid == 0.5
    assert oracle.tanh(np.float32([-1e4, 1e4, 0.0])).tolist() == [-1.0, 1.0, 0.0]


def test_sigmoid_table_is_platform_independent(oracle):
    # the 1024 x 4 coefficients are built from +,*,fma only; pin a few and the
    # whole table against float64 libm to 1 ulp
    tab = oracle.sigmoid_table()
    assert tab.shape == (1024, 4)
    z0 = (np.arange(1024) - 512) / 16.0
    assert np.abs(tab[:, 0] - _sig(z0)).max() < 6e-8
    assert tab[512].tolist() == [0.5, 0.015625, -3.970503481554033e-09, -5.080306436866522e-06]


def test_scaler_net_vs_float64(oracle, stages):
    # float32 canonical arithmetic vs float64 Keras equations on real heads
    for k in range(0, len(stages['scaler_in']), 5):
        head = stages['scaler_in'][k]
        got = oracle.scaler_forward(head)
        assert np.array_equal(got, stages['scaler_out'][k])      # fixture self-check
        assert np.abs(got - scaler_f64(head)).max() < 2e-4


def test_demux_net_vs_float64(oracle, stages):
    for k in range(0, len(stages['demux_in']), 4):
        win = stages['demux_in'][k]
        got = oracle.demux_forward(win)
        assert np.array_equal(got, stages['demux_out'][k])
        assert abs(got.sum() - 1) < 1e-6
        assert np.abs(got - demux_f64(win)).max() < 1e-4   # north_star tolerance


def test_lstm_layer_vs_torch(oracle):
    torch = pytest.importorskip('torch')
    m = load_model_arrays('MIN106-RNA001/demux-tetra-r4.npz')
    H = 48
    perm = np.concatenate([np.arange(0, H), np.arange(H, 2 * H),
                           np.arange(2 * H, 3 * H), np.arange(3 * H, 4 * H)])  # i,f,g,o
    lstm = torch.nn.LSTM(1, H, batch_first=True)
    with torch.no_grad():
        lstm.weight_ih_l0.copy_(torch.from_numpy(m['fwd_kernel'].T[perm].copy()))
        lstm.weight_hh_l0.copy_(torch.from_numpy(m['fwd_recurrent'].T[perm].copy()))
        lstm.bias_ih_l0.copy_(torch.from_numpy(m['fwd_bias'][perm].copy()))
        lstm.bias_hh_l0.zero_()
        rng = np.random.default_rng(0)
        x = rng.normal(0, 1, (300, 1)).astype(np.float32)
        x[:20] = -1000.0                                  # left padding
        want = lstm(torch.from_numpy(x)[None])[0][0].numpy()
    seq, last = oracle.lstm_layer('demux_fwd', x)
    assert np.abs(seq - want).max() < 2e-5
    assert np.array_equal(seq[-1], last)


def test_whole_networks_vs_torch_modules(oracle, stages):
    """poreplex_amd.torch_models (independent fp32 statement of the two Keras nets,
    incl. the Bidirectional merge and the left zero padding) vs the oracle."""
    torch = pytest.importorskip('torch')
    from poreplex_amd.torch_models import DemuxNet, ScalerNet
    with torch.no_grad():
        wins = np.ascontiguousarray(stages['demux_in'][:6], dtype=np.float32)
        want = np.exp(DemuxNet()(torch.from_numpy(wins)).numpy())
        got = np.stack([oracle.demux_forward(w) for w in wins])
        assert np.abs(got - want).max() < 1e-4            # north_star softmax tolerance
        assert np.array_equal(got.argmax(1), want.argmax(1))
        heads = np.ascontiguousarray(stages['scaler_in'][:4], dtype=np.float32)
        want = ScalerNet()(torch.from_numpy(heads)).numpy()
        got = np.stack([oracle.scaler_forward(h) for h in heads])
        assert np.abs(got - want).max() < 2e-4


def test_viterbi_vs_brute_force(oracle):
    """Exhaustive path enumeration for tiny T, both HMMs."""
    rng = np.random.default_rng(11)
    for which in (0, 1):
        hmm = oracle._hmm(which)
        S = hmm.n_states
        lt = np.full((S, S), -np.inf)
        for i in range(S):
            for j in range(S):
                if hmm.trans[i][j] > 0:
                    lt[i, j] = np.log(hmm.trans[i][j])
        lp = np.array([np.log(hmm.start_prob[s]) if hmm.start_prob[s] > 0 else -np.inf
                       for s in range(S)])
        for trial in range(6):
            T = 5
            x = rng.choice([71.5, 102, 112, 80, 65, 109, 95], T).astype(np.float32) \
                + rng.normal(0, 2, T).astype(np.float32)
            em = np.array([[oracle.emission(s, float(v), which) for s in range(S)] for v in x])
            best, arg = -np.inf, None
            for path in itertools.product(range(S), repeat=T):
                sc = lp[path[0]] + em[0, path[0]]
                for t in range(1, T):
                    sc += lt[path[t - 1], path[t]] + em[t, path[t]]
                if sc > best:
                    best, arg = sc, path
            logp, path = oracle.viterbi(x, which)
            assert tuple(path) == arg
            assert abs(logp - best) < 1e-9


def test_emission_formula(oracle):
    # pomegranate NormalDistribution / GeneralMixtureModel log-density
    from scipy.stats import norm
    from scipy.special import logsumexp
    hmm = oracle._hmm(0)
    for s in range(hmm.n_states):
        for x in (40.0, 80.5, 110.0, 160.0):
            w = np.array([hmm.mix_weight[s][k] for k in range(hmm.n_mix[s])])
            want = logsumexp([np.log(w[k] / w.sum()) +
                              norm.logpdf(x, hmm.mix_mu[s][k], hmm.mix_sigma[s][k])
                              for k in range(hmm.n_mix[s])])
            assert abs(oracle.emission(s, x) - want) < 1e-9


def test_segments_last_run_wins(oracle):
    path = np.int32([4, 4, 1, 1, 1, 3, 3, 1, 5])
    first, last = oracle.segments(path)
    assert (first[1], last[1]) == (7, 7)       # repeated state keeps its LAST run
    assert (first[4], last[4]) == (0, 1) and (first[3], last[3]) == (5, 6)
    assert first[0] == -1 and first[2] == -1
