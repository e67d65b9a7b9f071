#!/usr/bin/env python
"""Decision flips of the two recurrent networks' arithmetics, measured on the GPU (run on an MI355X box).

a4 / a12 hand their networks to TensorFlow in the reference (signal_loader.py:96-97, barcoding.py:106-107):
float32 Keras LSTM equations in an unspecified summation order.  This build has two canonical evaluations
(include/pxg.h pxg_lstm_arith: q8 = exact fixed point on the int8 matrix pipe, f32 = fma chains) whose
kernels are bit-exact with their oracle restatements (tests -m gpu) -- what no test can show without
TensorFlow is whether either takes the DECISIONS the reference would take.  This tool counts decisions
between three evaluations of the same reads:
    q8, f32   the two product arithmetics, whole pipeline on the GPU;
    f64       the exact Keras equations in float64 (torch on the GPU, libm-grade exp / tanh), outputs rounded
              to float32 where Keras hands float32 to the NumPy glue; everything between and behind the two
              networks (pooling, scaling, Viterbi, window rules, robust z-score) is the GPU pipeline run
              with f64's own (scale, shift) INJECTED -- those stages are integer / defined float arithmetic
              pinned by the reference's goldens, identical for every arithmetic.
on four read sets:
    bench        reads of bench.py's generator (>= 20 000),
    adversarial  the same generator with levels midway between neighbouring states' means, per-read drift and
                 adapters at the 260 / 3 000 gates of BarcodeDemultiplexer.push,
    qc-edge      heads scaled / shifted by bisection until the scaler's output sits within 1e-4 of one of the
                 four scaling-QC bounds (signal_loader.py:65-73),
    call-edge    classifier windows blended between a barcode prototype and noise by bisection until the
                 softmax maximum sits within 1e-3 of the calling threshold 0.97972751 (barcoding.py:41-45).
Prints one JSON document (committed as profiles/r04/decision_flips_gpu.json; bench.py quotes it).
usage: python tools/decision_flips_gpu.py [--reads 20480] [--adv-reads 8192] [--edge 2048]
"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (before libpxg.so: both must bind the same HIP runtime)

torch.zeros(1, device='cuda')
from poreplex_amd import native as N  # noqa: E402
from poreplex_amd.config import default_config, load_model_arrays  # noqa: E402
from poreplex_amd.synth import load_prototypes, synth_batch  # noqa: E402

DEV = torch.device('cuda')
F64 = torch.float64


def _t(a):
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=DEV)


def _cell(z, c, H):
    i, f, g, o = torch.sigmoid(z[:, :H]), torch.sigmoid(z[:, H:2 * H]), torch.tanh(z[:, 2 * H:3 * H]), torch.sigmoid(z[:, 3 * H:])
    c = f * c + i * g
    return o * torch.tanh(c), c


def scaler_f64(heads):
    """[B, T] float32 -> [B, 2] float64: LSTM(48, seq) -> LSTM(48) -> Dense(2), Keras equations."""
    m = {k: _t(v) for k, v in load_model_arrays('MIN106-RNA001/scaler-r3.npz').items() if np.ndim(v) > 0 and hasattr(v, 'shape')}
    out = []
    for a in range(0, len(heads), 8192):
        x = _t(heads[a:a + 8192])
        B, T = x.shape
        h1, c1, h2, c2 = (torch.zeros(B, 48, dtype=F64, device=DEV) for _ in range(4))
        for t in range(T):
            h1, c1 = _cell(x[:, t:t + 1] * m['lstm1_kernel'] + h1 @ m['lstm1_recurrent'] + m['lstm1_bias'], c1, 48)
            h2, c2 = _cell(h1 @ m['lstm2_kernel'] + h2 @ m['lstm2_recurrent'] + m['lstm2_bias'], c2, 48)
        out.append((h2 @ m['dense_kernel'] + m['dense_bias']).cpu().numpy())
    return np.concatenate(out) if out else np.zeros((0, 2))


def demux_f64(wins):
    """[B, 300] float32 -> softmax [B, 5] float64."""
    m = {k: _t(v) for k, v in load_model_arrays('MIN106-RNA001/demux-tetra-r4.npz').items() if np.ndim(v) > 0 and hasattr(v, 'shape')}
    out = []
    for a in range(0, len(wins), 8192):
        x = _t(wins[a:a + 8192])
        B, T = x.shape
        seq = torch.zeros(B, T, 96, dtype=F64, device=DEV)
        for name, lo, order in (('fwd', 0, range(T)), ('bwd', 48, range(T - 1, -1, -1))):
            h, c = torch.zeros(B, 48, dtype=F64, device=DEV), torch.zeros(B, 48, dtype=F64, device=DEV)
            for t in order:
                h, c = _cell(x[:, t:t + 1] * m[name + '_kernel'] + h @ m[name + '_recurrent'] + m[name + '_bias'], c, 48)
                seq[:, t, lo:lo + 48] = h
        h, c = torch.zeros(B, 64, dtype=F64, device=DEV), torch.zeros(B, 64, dtype=F64, device=DEV)
        for t in range(T):
            h, c = _cell(seq[:, t] @ m['top_kernel'] + h @ m['top_recurrent'] + m['top_bias'], c, 64)
        out.append(torch.softmax(h @ m['dense_kernel'] + m['dense_bias'], dim=1).cpu().numpy())
    return np.concatenate(out) if out else np.zeros((0, 5))


def destandardise(cfg, pred32):
    """signal_loader.py:98-102 under NumPy-1.x casting: float32 Horner, inclusive float32 bounds."""
    xf = cfg.scaler_xfrm
    scale = np.float32(xf[1]) * pred32[:, 0] + np.float32(xf[0])
    shift = np.float32(xf[3]) * pred32[:, 1] + np.float32(xf[2])
    ok = ((scale >= np.float32(cfg.scaler_qc_scale[0])) & (scale <= np.float32(cfg.scaler_qc_scale[1])) &
          (shift >= np.float32(cfg.scaler_qc_shift[0])) & (shift <= np.float32(cfg.scaler_qc_shift[1])))
    return scale.astype(np.float32), shift.astype(np.float32), ok


def call_rule(cfg, probs):
    """barcoding.py:108-118: label = argmax - decoys; called iff label >= 0 and float32 score >= threshold."""
    label = probs.argmax(1) - int(cfg.number_of_decoy_labels)
    return label, (label >= 0) & (probs.max(1).astype(np.float32) >= np.float64(cfg.score_threshold))


class Pipelines:
    def __init__(self):
        self.ctx = {}
        for arith in ('q8', 'f32'):
            cfg = default_config()
            cfg['signal_processing']['lstm_arith'] = arith
            os.environ.pop('PXG_LSTM_ARITH', None)
            self.ctx[arith] = N.NativeContext(cfg, device_id=0)
        self.cfg = self.ctx['q8'].cfg

    def product(self, arith, b):
        c = self.ctx[arith]
        c.upload(b['arena'], b['offsets'], b['calib'])
        c.run(N.STAGE_ALL_DEMUX)
        return c.download().copy()

    def exact(self, b):
        """f64 networks around the GPU's own pooling / Viterbi / window stages."""
        c = self.ctx['f32']
        heads, st = c.head_pool(b['arena'], b['offsets'], b['calib'])
        scored = st == 0
        pred32 = np.zeros((len(st), 2), dtype=np.float32)
        pred32[scored] = scaler_f64(heads[scored]).astype(np.float32)
        scale, shift, ok = destandardise(self.cfg, pred32)
        inject = np.stack([np.where(ok & scored, scale, 1.0), np.where(ok & scored, shift, 0.0)], axis=1).astype(np.float32)
        c.upload(b['arena'], b['offsets'], b['calib'], inject)
        c.run(N.STAGE_ALL_DEMUX)
        rec = c.download().copy()
        wins = c.download_windows()
        pushed = (rec['bc_pushed'] == 1) & ok & scored
        p64 = np.zeros((len(st), 5))
        p64[pushed] = demux_f64(wins[pushed])
        return {'scored': scored, 'qc_ok': ok & scored, 'scale': scale, 'shift': shift, 'pred': pred32, 'rec': rec,
                'pushed': pushed, 'p64': p64}

    def close(self):
        for c in self.ctx.values():
            c.close()


def compare_products(a, b):
    both = (a['status'] == 0) & (b['status'] == 0)
    pushed = both & (a['bc_pushed'] == 1) & (b['bc_pushed'] == 1)
    return {
        'reads': int(len(a)), 'status_flips': int((a['status'] != b['status']).sum()),
        'reads_with_a_segment_boundary_moved': int(((a['seg_first'] != b['seg_first']) | (a['seg_last'] != b['seg_last'])).any(1)[both].sum()),
        'window_gate_flips': int((a['bc_pushed'] != b['bc_pushed'])[both].sum()),
        'windows_compared': int(pushed.sum()),
        'argmax_flips': int((a['bc_label'] != b['bc_label'])[pushed].sum()),
        'called_uncalled_flips': int((a['bc_called'] != b['bc_called'])[pushed].sum()),
        'phred_changes': int((a['bc_phred'] != b['bc_phred'])[pushed].sum()),
        'scaler_pred_max_abs_diff': np.abs(a['scaler_pred'][both] - b['scaler_pred'][both]).max(axis=0).tolist() if both.any() else None,
        'softmax_max_abs_diff': float(np.abs(a['probs'][pushed] - b['probs'][pushed]).max()) if pushed.any() else None,
    }


def compare_exact(cfg, prod, ex):
    """A product arithmetic's whole pipeline against the f64 pipeline."""
    TOO_SHORT, QC_FAIL = N.STATUS_CODE['scaler_signal_too_short'], N.STATUS_CODE['scaling_qc_fail']
    scored = ex['scored']
    assert np.array_equal(scored, prod['status'] != TOO_SHORT)
    p_ok = scored & (prod['status'] != QC_FAIL)
    both = p_ok & ex['qc_ok']
    r = ex['rec']
    moved = ((prod['seg_first'] != r['seg_first']) | (prod['seg_last'] != r['seg_last'])).any(1)
    adapter = int(cfg.segmentation_model.adapter_state)
    found_flip = (prod['seg_first'][:, adapter] >= 0) != (r['seg_first'][:, adapter] >= 0)
    pushed = both & (prod['bc_pushed'] == 1) & ex['pushed']
    lab64, call64 = call_rule(cfg, ex['p64'])
    return {
        'reads': int(len(prod)), 'reads_scored_by_scaler': int(scored.sum()),
        'scaler_pred_max_abs_diff': np.abs(prod['scaler_pred'][scored].astype(np.float64) - ex['pred'][scored]).max(axis=0).tolist(),
        'scale_max_abs_diff': float(np.abs(prod['scale'][both].astype(np.float64) - ex['scale'][both]).max()),
        'shift_max_abs_diff': float(np.abs(prod['shift'][both].astype(np.float64) - ex['shift'][both]).max()),
        'scaling_qc_flips': int((p_ok != ex['qc_ok'])[scored].sum()),
        'reads_segmented_by_both': int(both.sum()),
        'reads_with_a_segment_boundary_moved': int(moved[both].sum()),
        'adapter_found_flips': int(found_flip[both].sum()),
        'window_gate_flips': int(((prod['bc_pushed'] == 1) != ex['pushed'])[both & ~found_flip].sum()),
        'windows_compared': int(pushed.sum()),
        'softmax_max_abs_diff_whole_pipeline': float(np.abs(prod['probs'][pushed][:, :5] - ex['p64'][pushed]).max()) if pushed.any() else None,
        'argmax_flips': int(((prod['bc_label'][pushed]) != lab64[pushed]).sum()),
        'called_uncalled_flips_at_threshold': int(((prod['bc_called'][pushed] == 1) != call64[pushed]).sum()),
        'barcodes_called': int((prod['bc_called'][pushed] == 1).sum()),
    }


def read_set(P, name, n, seed, **kw):
    out = {'set': name, 'reads': n, 'seed': seed, 'generator': kw or 'bench.py default'}
    recs = {'q8': [], 'f32': []}
    ex_all = []
    for a in range(0, n, 4096):          # 4 096 reads of ~60 000 samples = 0.5 GB per chunk on the host
        b = synth_batch(min(4096, n - a), seed=seed + a, samples_per_read=60000, short_fraction=0.01, **kw)
        for arith in recs:
            recs[arith].append(P.product(arith, b))
        ex_all.append(P.exact(b))
    q8, f32 = np.concatenate(recs['q8']), np.concatenate(recs['f32'])
    ex = {k: np.concatenate([e[k] for e in ex_all]) for k in ex_all[0]}
    out['q8_vs_f32'] = compare_products(q8, f32)
    out['q8_vs_f64'] = compare_exact(P.cfg, q8, ex)
    out['f32_vs_f64'] = compare_exact(P.cfg, f32, ex)
    return out


def qc_edge(P, n, seed):
    """Heads whose q8 scaler output lies within 1e-4 of a QC bound, by bisection on the head's amplitude
    (scale bounds) or offset (shift bounds); then the QC verdict of the three evaluations."""
    cfg = P.cfg
    b = synth_batch(n, seed=seed, samples_per_read=32000, short_fraction=0.0)
    heads, st = P.ctx['q8'].head_pool(b['arena'], b['offsets'], b['calib'])
    heads = heads[st == 0]
    n = len(heads)
    xf = cfg.scaler_xfrm
    which = np.arange(n) % 4                                  # scale lo / hi, shift lo / hi
    bound = np.array([cfg.scaler_qc_scale[0], cfg.scaler_qc_scale[1], cfg.scaler_qc_shift[0], cfg.scaler_qc_shift[1]])[which]
    # aim for a point spread evenly over [bound - 1e-4, bound + 1e-4], not for the bound itself
    target = bound + np.random.default_rng(seed).uniform(-1e-4, 1e-4, n)
    is_scale = which < 2

    def apply(p):
        return np.where(is_scale[:, None], heads * p[:, None], np.where(heads != 0, heads + p[:, None], 0.0)).astype(np.float32)

    def value(h):
        pred = P.ctx['q8'].scaler_lstm(h)
        scale, shift, _ = destandardise(cfg, pred)
        return np.where(is_scale, scale, shift).astype(np.float64)
    lo = np.where(is_scale, 0.4, -60.0)
    hi = np.where(is_scale, 2.5, 60.0)
    v_lo, v_hi = value(apply(lo)), value(apply(hi))
    bracket = (v_lo - target) * (v_hi - target) < 0
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        v = value(apply(mid))
        same = (v - target) * (v_lo - target) > 0
        lo, v_lo = np.where(same, mid, lo), np.where(same, v, v_lo)
        hi = np.where(same, hi, mid)
    final = apply(0.5 * (lo + hi))
    v = value(final)
    near = bracket & (np.abs(v - bound) <= 1e-4)
    final = final[near]
    verdict = {}
    for arith in ('q8', 'f32'):
        verdict[arith] = destandardise(cfg, P.ctx[arith].scaler_lstm(final))[2]
    verdict['f64'] = destandardise(cfg, scaler_f64(final).astype(np.float32))[2]
    dist = np.abs(v[near] - bound[near])
    return {'set': 'qc-edge', 'heads_tried': int(n), 'heads_within_1e-4_of_a_bound': int(near.sum()),
            'median_distance_to_bound': float(np.median(dist)) if near.any() else None,
            'heads_within_1e-6': int((dist <= 1e-6).sum()),
            'qc_verdict_flips': {'q8_vs_f32': int((verdict['q8'] != verdict['f32']).sum()),
                                 'q8_vs_f64': int((verdict['q8'] != verdict['f64']).sum()),
                                 'f32_vs_f64': int((verdict['f32'] != verdict['f64']).sum())}}


def call_edge(P, n, seed):
    """Windows whose q8 softmax maximum lies within 1e-3 of the calling threshold: a barcode prototype blended
    with noise, blend found by bisection; then the call of the three evaluations."""
    cfg = P.cfg
    rng = np.random.default_rng(seed)
    proto = load_prototypes()
    cls = rng.integers(1, 5, n)
    noise = rng.normal(0, 1, (n, 300)).astype(np.float32)
    thr = float(cfg.score_threshold)
    aim = thr + rng.uniform(-1e-3, 1e-3, n)          # spread over the band, not piled on the threshold

    def window(lam):
        return (lam[:, None] * proto[cls] + (1 - lam[:, None]) * noise).astype(np.float32)

    def score(w):
        return P.ctx['q8'].demux_lstm(w)[:, :5].max(1).astype(np.float64)
    lo, hi = np.zeros(n), np.ones(n)
    s_lo, s_hi = score(window(lo)), score(window(hi))
    bracket = (s_lo < aim) & (s_hi > aim)
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        below = score(window(mid)) < aim
        lo, hi = np.where(below, mid, lo), np.where(below, hi, mid)
    w = window(0.5 * (lo + hi))
    s = score(w)
    near = bracket & (np.abs(s - thr) <= 1e-3)
    w = w[near]
    res = {}
    for arith in ('q8', 'f32'):
        res[arith] = call_rule(cfg, P.ctx[arith].demux_lstm(w)[:, :5].astype(np.float64))
    res['f64'] = call_rule(cfg, demux_f64(w))
    dist = np.abs(s[near] - thr)

    def flips(a, b):
        return {'argmax': int((res[a][0] != res[b][0]).sum()), 'called_uncalled': int((res[a][1] != res[b][1]).sum())}
    return {'set': 'call-edge', 'windows_tried': int(n), 'windows_within_1e-3_of_the_threshold': int(near.sum()),
            'median_distance_to_threshold': float(np.median(dist)) if near.any() else None,
            'windows_within_1e-6': int((dist <= 1e-6).sum()),
            'flips': {'q8_vs_f32': flips('q8', 'f32'), 'q8_vs_f64': flips('q8', 'f64'), 'f32_vs_f64': flips('f32', 'f64')}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=20480)
    ap.add_argument('--adv-reads', type=int, default=8192)
    ap.add_argument('--edge', type=int, default=2048)
    ap.add_argument('--seed', type=int, default=924)
    args = ap.parse_args()
    t0 = time.time()
    P = Pipelines()
    out = {'device': P.ctx['q8'].device_info()['name'], 'sets': []}
    out['sets'].append(read_set(P, 'bench', args.reads, args.seed))
    out['sets'].append(read_set(P, 'adversarial', args.adv_reads, args.seed + 500000,
                                adversarial={'blend': 0.5, 'drift': 4.0, 'adapter_gate': 0.5}))
    out['sets'].append(qc_edge(P, args.edge, args.seed + 900000))
    out['sets'].append(call_edge(P, args.edge, args.seed + 950000))
    P.close()
    out['wall_s'] = round(time.time() - t0, 1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
