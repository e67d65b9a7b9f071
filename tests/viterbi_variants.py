"""Formula variants of the pomegranate Viterbi (rows a6 / a7 / a19: parity unpinned).

pomegranate is not installable here, so oracle/pxo_core.c restates its Viterbi from the
published algorithm.  "GPU == oracle" therefore proves the kernels, not the
restatement.  This module is the other half: a batched NumPy Viterbi with every choice a
DIFFERENT-BUT-LEGAL implementation of the same model is free to make exposed as a knob, so
tests/test_decision_flips.py can count how many DECISIONS (segment boundaries, in-read adapter
candidates) each choice moves on bench-sized samples.

Knobs (the canonical value first = what oracle/pxo_core.c:366-470 and the kernels do):
  mixture   'pairwise'   lp = pair_lse(lp, logpdf_k + log w_k), components in order
            'logsumexp'  m + log(sum_k exp(l_k - m)),  m = max_k l_k
            'logsum'     log(sum_k w_k * pdf_k) with pdf_k = exp(logpdf_k)
  logpdf    'pomegranate'  -log(sigma * 2.50662827463) - d*d * (1 / (2 sigma^2))
            'textbook'     -0.5 log(2 pi) - log sigma - 0.5 (d / sigma)^2
  trans     'raw'        log of the configured probabilities
            'bake'       p / sum_row(p) first (HiddenMarkovModel.bake's normalisation)
  order     'sorted'     in-edges scanned / terminal state picked in name-sorted order
            'config'     ... in the order the preset lists the states
  dtype     float64 | longdouble
Test infrastructure: imported only from tests/.
"""
import numpy as np

CANONICAL = dict(mixture='pairwise', logpdf='pomegranate', trans='raw', order='sorted',
                 dtype=np.float64)

VARIANTS = (
    ('canonical (NumPy restatement of the oracle)', {}),
    ('mixture = logsumexp over components', {'mixture': 'logsumexp'}),
    ('mixture = log(sum w*pdf)', {'mixture': 'logsum'}),
    ('normal log-pdf in textbook form', {'logpdf': 'textbook'}),
    ('transitions renormalised as bake() does', {'trans': 'bake'}),
    ('in-edge / terminal order = preset order', {'order': 'config'}),
    ('all of the above at once', {'mixture': 'logsumexp', 'logpdf': 'textbook', 'trans': 'bake',
                                  'order': 'config'}),
    ('canonical in longdouble', {'dtype': np.longdouble}),
    ('all of the above in longdouble', {'mixture': 'logsum', 'logpdf': 'textbook', 'trans': 'bake',
                                        'order': 'config', 'dtype': np.longdouble}),
)


class Model:
    """Tables of one pxg_hmm under one set of knobs."""

    def __init__(self, hmm, **knobs):
        k = dict(CANONICAL, **knobs)
        self.k, dt = k, k['dtype']
        S = self.S = int(hmm.n_states)
        self.dt = dt
        tr = np.array([[hmm.trans[i][j] for j in range(S)] for i in range(S)], dtype=dt)
        if k['trans'] == 'bake':
            tr = tr / tr.sum(axis=1, keepdims=True)
        with np.errstate(divide='ignore'):
            self.logtr = np.log(tr)
            self.logstart = np.log(np.array([hmm.start_prob[i] for i in range(S)], dtype=dt))
        rank = [int(hmm.name_rank[i]) for i in range(S)]
        self.scan = np.argsort(rank) if k['order'] == 'sorted' else np.arange(S)
        self.comp = []
        for s in range(S):
            n = int(hmm.n_mix[s])
            mu = np.array([hmm.mix_mu[s][j] for j in range(n)], dtype=dt)
            sd = np.array([hmm.mix_sigma[s][j] for j in range(n)], dtype=dt)
            w = np.array([hmm.mix_weight[s][j] for j in range(n)], dtype=dt)
            self.comp.append((mu, sd, w / w.sum()))

    def _logpdf(self, x, mu, sd):
        dt = self.dt
        d = x - mu
        if self.k['logpdf'] == 'pomegranate':
            return -np.log(sd * dt(2.50662827463)) - (d * d) * (dt(1) / (dt(2) * sd * sd))
        z = d / sd
        two_pi = dt(2) * np.arccos(dt(-1))              # pi in the working precision
        return -dt(0.5) * np.log(two_pi) - np.log(sd) - dt(0.5) * z * z

    def emissions(self, x):
        """x [..., T] float32 -> e [..., T, S] in the model's dtype."""
        dt = self.dt
        x = np.asarray(x).astype(dt)
        out = np.empty(x.shape + (self.S,), dtype=dt)
        for s, (mu, sd, w) in enumerate(self.comp):
            if len(mu) == 1:
                out[..., s] = self._logpdf(x, mu[0], sd[0])
                continue
            parts = [self._logpdf(x, mu[j], sd[j]) for j in range(len(mu))]
            with np.errstate(divide='ignore', over='ignore', invalid='ignore'):
                if self.k['mixture'] == 'pairwise':
                    lp = np.full(x.shape, -np.inf, dtype=dt)
                    for j, l in enumerate(parts):
                        t = l + np.log(w[j])
                        hi, lo = np.maximum(lp, t), np.minimum(lp, t)
                        both = hi + np.log(np.exp(lo - hi) + dt(1))
                        lp = np.where(np.isneginf(lo), hi, both)
                elif self.k['mixture'] == 'logsumexp':
                    stack = np.stack([l + np.log(w[j]) for j, l in enumerate(parts)], axis=-1)
                    m = stack.max(axis=-1)
                    lp = m + np.log(np.exp(stack - m[..., None]).sum(axis=-1))
                else:
                    lp = np.log(sum(w[j] * np.exp(l) for j, l in enumerate(parts)))
            out[..., s] = lp
        return out


def viterbi_batch(model, x, lengths):
    """x [B, Tmax] float32 (rows padded with anything), lengths [B] -> (paths [B, Tmax] int8
    valid up to lengths[b], logp [B]).  pomegranate semantics: v0 = log pi + e; strict '>'
    over the in-edges in scan order (first maximum wins = argmax over the reordered axis);
    first maximum of the last column in scan order; traceback."""
    B, T = x.shape
    S, scan = model.S, model.scan
    lengths = np.asarray(lengths, dtype=np.int64)
    e = model.emissions(x)
    logtr_scan = model.logtr[scan]                      # [k in scan order, s]
    bp = np.empty((T, B, S), dtype=np.int8)
    with np.errstate(invalid='ignore'):
        v = model.logstart[None, :] + e[:, 0, :]
        ident = np.broadcast_to(np.arange(S, dtype=np.int8), (B, S))
        bp[0] = ident
        for t in range(1, T):
            cand = v[:, scan, None] + logtr_scan[None]  # [B, k, s]
            arg = cand.argmax(axis=1)                   # first maximum in scan order
            best = np.take_along_axis(cand, arg[:, None, :], axis=1)[:, 0, :]
            nv = best + e[:, t, :]
            live = (t < lengths)[:, None]
            v = np.where(live, nv, v)
            bp[t] = np.where(live, scan[arg].astype(np.int8), ident)
    end = scan[v[:, scan].argmax(axis=1)]
    logp = v[np.arange(B), end]
    paths = np.empty((B, T), dtype=np.int8)
    s = end.astype(np.int64)
    rows = np.arange(B)
    for t in range(T - 1, -1, -1):
        paths[:, t] = s
        s = bp[t][rows, s].astype(np.int64)
    return paths, logp


def runs_to_segments(path, n_slots):
    """signal_analyzer.py:354-362: name -> (first, last), a later run overwrites."""
    first = np.full(n_slots, -1, dtype=np.int32)
    last = np.full(n_slots, -1, dtype=np.int32)
    if len(path):
        cut = np.nonzero(np.diff(path))[0] + 1
        starts = np.r_[0, cut]
        ends = np.r_[cut - 1, len(path) - 1]
        for a, b in zip(starts.tolist(), ends.tolist()):
            first[path[a]], last[path[a]] = a, b
    return first, last


def unsplit_windows(cfg, n_events, first, stride, payload_start, rate):
    """Window frame of signal_analyzer.py:384-388 (as oracle/pxo_unsplit.c:61-87 restates it):
    list of (k0, k1) inclusive event ranges."""
    size, step = int(cfg.unsplit_window_size * rate), int(cfg.unsplit_window_step * rate)
    out = []
    if n_events <= 0 or step <= 0:
        return out
    last_end = first + stride * (n_events - 1) + 1
    left = payload_start
    while left < last_end:
        k0 = 0 if left - first <= 0 else (left - first + stride - 1) // stride
        k1 = -1 if left + size - first < 0 else (left + size - first) // stride
        k1 = min(k1, n_events - 1)
        if k1 < k0:
            break
        out.append((int(k0), int(k1)))
        left += step
    return out


def unsplit_candidates(cfg, path, k0, n_events, first, stride, payload_start, rate):
    """Run analysis of one window's path (signal_analyzer.py:393-418)."""
    hmm = cfg.unsplit_model
    A, LL, LH = hmm.adapter_state, hmm.leader_low_state, hmm.leader_high_state
    strict_duration = int(cfg.unsplit_strict_duration * rate)
    cut_total = (int(cfg.unsplit_loosen_full_length * rate), int(cfg.unsplit_strict_full_length * rate))
    cut_adapter = (int(cfg.unsplit_loosen_dna_length * rate), int(cfg.unsplit_strict_dna_length * rate))
    out, leader = [], -1
    cut = np.nonzero(np.diff(path))[0] + 1
    for t, e in zip(np.r_[0, cut].tolist(), np.r_[cut - 1, len(path) - 1].tolist()):
        st = int(path[t])
        if st not in (A, LL, LH):
            leader = -1
            continue
        if leader < 0:
            leader = t
        if st == A:
            ev_last, ev_lead, ev_first = k0 + e, k0 + leader, k0 + t
            adapter_end = first + stride * ev_last + 1 if ev_last == n_events - 1 \
                else first + stride * (ev_last + 1)
            lead_at = first + stride * ev_lead
            strict = int(lead_at - payload_start <= strict_duration)
            if adapter_end - lead_at >= cut_total[strict] and \
                    adapter_end - (first + stride * ev_first) >= cut_adapter[strict]:
                out.append([int(lead_at), int(1 + adapter_end)])
            leader = -1
    return out
