"""ctypes front-end of the ORACLE (oracle/_build/libpxo.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from poreplex_amd/.  It borrows the POD
layouts (ctypes Structures / NumPy dtypes of include/pxg.h) from
poreplex_amd.native so oracle results compare field by field with the HIP path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from poreplex_amd import native as N

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, '_build', 'libpxo.so')
REF_LIB_PATH = os.path.join(HERE, '_ref', 'libscrappie_ref.so')

_lib = None
SPIKE_ROWS = 64        # spike rows per read of the first try (every spike is counted: rerun if short)


def build(force=False):
    """Compile the C restatement (and, when /root/reference exists, the
    reference's own event detector into oracle/_ref/)."""
    if force or not os.path.isfile(LIB_PATH) or _stale():
        subprocess.check_call(['make', '-C', HERE, '--no-print-directory'],
                              stdout=subprocess.DEVNULL)
    return LIB_PATH


def _stale():
    try:
        t = os.path.getmtime(LIB_PATH)
        srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(('.c', '.h'))]
        srcs.append(os.path.join(HERE, '..', 'include', 'pxg.h'))
        return any(os.path.getmtime(s) > t for s in srcs)
    except OSError:
        return True


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        vp, i64, i32, f32, f64 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_double
        cfgp = C.POINTER(N.PxgConfig)
        sig = {
            'pxo_raw_to_pa': (None, [vp, i64, vp, vp]),
            'pxo_np_sum_f32': (f32, [vp, i64]),
            'pxo_head_pool': (i32, [vp, i64, vp, i32, i32, i32, vp]),
            'pxo_pool_scale': (i64, [vp, i64, vp, i32, f32, f32, vp]),
            'pxo_sigmoid_table': (None, [vp]),
            'pxo_expf': (f32, [f32]), 'pxo_sigmoid': (f32, [f32]), 'pxo_tanh': (f32, [f32]),
            'pxo_scaler_forward': (None, [cfgp, vp, i32, vp]),
            'pxo_scaler_transform': (i32, [cfgp, vp, vp]),
            'pxo_demux_forward': (None, [cfgp, vp, i32, vp]),
            'pxo_lstm_layer': (None, [C.POINTER(N.PxgLstmLayer), vp, i32, i32, vp, vp]),
            'pxo_viterbi': (f64, [C.POINTER(N.PxgHmm), vp, i32, vp]),
            'pxo_hmm_emission': (f64, [C.POINTER(N.PxgHmm), i32, f64]),
            'pxo_segments': (None, [vp, i32, vp, vp]),
            'pxo_normalize_signal': (None, [vp, i32, vp]),
            'pxo_barcode_window': (i32, [cfgp, vp, i32, vp]),
            'pxo_phred': (i32, [cfgp, f32]),
            'pxo_barcode_call': (None, [cfgp, vp, vp]),
            'pxo_detect_events': (i64, [vp, i64, i64, i64, f32, f32, f32, vp, i64]),
            'pxo_medfilt': (None, [vp, i64, i32, vp]),
            'pxo_polya': (None, [cfgp, vp, i64, i32, i32, f64, vp, vp, i32]),
            'pxo_best_polya_interval': (i32, [cfgp, vp, vp, i32, vp, vp]),
            'pxo_guppy_event_means': (i32, [vp, i64, vp, i64, i64, i32, f32, f32, vp, vp]),
            'pxo_guppy_event_table': (i32, [vp, i64, vp, i64, i64, i32, f32, f32, vp, vp, vp]),
            'pxo_unsplit_scan': (i32, [cfgp, vp, i64, i64, i32, i64, f64, vp, i32]),
            'pxo_unsplit_scan_events': (i32, [cfgp, vp, vp, i64, i64, f64, vp, i32]),
            'pxo_process_read': (None, [cfgp, vp, i64, vp, vp, C.c_uint32, vp, vp, i32]),
            'pxo_process_batch': (None, [cfgp, i64, vp, vp, vp, vp, C.c_uint32, vp, vp, i32]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _cal(calib_row):
    return np.ascontiguousarray(calib_row, dtype=N.CALIB_DTYPE).reshape(1)


class Oracle:
    """CPU restatement bound to one config (same dict the product takes)."""

    def __init__(self, config):
        self.ncfg = N.NativeConfig(config)
        self.cfg = self.ncfg.struct
        self.state_names = self.ncfg.state_names
        self.L = lib()

    # ---- signal ----------------------------------------------------------
    def raw_to_pa(self, raw, calib_row):
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        out = np.empty(len(raw), dtype=np.float32)
        self.L.pxo_raw_to_pa(_p(raw), len(raw), _p(_cal(calib_row)), _p(out))
        return out

    def np_sum_f32(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        return np.float32(self.L.pxo_np_sum_f32(_p(a), len(a)))

    def head_pool(self, raw, calib_row):
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        out = np.zeros(self.cfg.scaler_length // self.cfg.stride, dtype=np.float32)
        st = self.L.pxo_head_pool(_p(raw), len(raw), _p(_cal(calib_row)),
                                  self.cfg.scaler_length, self.cfg.stride,
                                  self.cfg.scaler_min_length, _p(out))
        return out, st

    def pool_scale(self, raw, calib_row, scale, shift):
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        out = np.zeros(len(raw) // self.cfg.stride, dtype=np.float32)
        self.L.pxo_pool_scale(_p(raw), len(raw), _p(_cal(calib_row)), self.cfg.stride,
                              np.float32(scale), np.float32(shift), _p(out))
        return out

    # ---- nets ------------------------------------------------------------
    def expf(self, x):
        return np.array([self.L.pxo_expf(float(v)) for v in np.atleast_1d(x)], np.float32)

    def sigmoid(self, x):
        return np.array([self.L.pxo_sigmoid(float(v)) for v in np.atleast_1d(x)], np.float32)

    def tanh(self, x):
        return np.array([self.L.pxo_tanh(float(v)) for v in np.atleast_1d(x)], np.float32)

    def sigmoid_table(self):
        tab = np.zeros((1024, 4), dtype=np.float32)
        self.L.pxo_sigmoid_table(_p(tab))
        return tab

    def scaler_forward(self, head):
        head = np.ascontiguousarray(head, dtype=np.float32)
        pred = np.zeros(2, dtype=np.float32)
        self.L.pxo_scaler_forward(C.byref(self.cfg), _p(head), len(head), _p(pred))
        return pred

    def scaler_transform(self, pred):
        pred = np.ascontiguousarray(pred, dtype=np.float32)
        ss = np.zeros(2, dtype=np.float32)
        st = self.L.pxo_scaler_transform(C.byref(self.cfg), _p(pred), _p(ss))
        return ss, st

    def demux_forward(self, win):
        win = np.ascontiguousarray(win, dtype=np.float32)
        probs = np.zeros(self.cfg.demux_dense.out_dim, dtype=np.float32)
        self.L.pxo_demux_forward(C.byref(self.cfg), _p(win), len(win), _p(probs))
        return probs

    def lstm_layer(self, which, x, reverse=False):
        layer = getattr(self.cfg, which)
        x = np.ascontiguousarray(x, dtype=np.float32)
        T = x.shape[0]
        seq = np.zeros((T, layer.units), dtype=np.float32)
        last = np.zeros(layer.units, dtype=np.float32)
        self.L.pxo_lstm_layer(C.byref(layer), _p(x), T, int(reverse), _p(seq), _p(last))
        return seq, last

    # ---- HMM -------------------------------------------------------------
    def _hmm(self, which):
        return self.cfg.unsplit_model if which else self.cfg.segmentation_model

    def viterbi(self, signal, which_model=0):
        x = np.ascontiguousarray(signal, dtype=np.float32)
        path = np.zeros(len(x), dtype=np.int32)
        logp = self.L.pxo_viterbi(C.byref(self._hmm(which_model)), _p(x), len(x), _p(path))
        return logp, path

    def emission(self, state, x, which_model=0):
        return self.L.pxo_hmm_emission(C.byref(self._hmm(which_model)), state, float(x))

    def segments(self, path):
        path = np.ascontiguousarray(path, dtype=np.int32)
        first = np.zeros(N.PXG_N_SEGMENTS, dtype=np.int32)
        last = np.zeros(N.PXG_N_SEGMENTS, dtype=np.int32)
        self.L.pxo_segments(_p(path), len(path), _p(first), _p(last))
        return first, last

    def detect_segments(self, signal, which_model=0):
        """name -> (first, last), like SignalAnalysis.detect_segments."""
        scan = self.cfg.segmentation_scan_limit // self.cfg.stride
        _, path = self.viterbi(np.asarray(signal)[:scan], which_model)
        first, last = self.segments(path)
        return {name: (int(first[i]), int(last[i]))
                for i, name in enumerate(self.state_names) if first[i] >= 0}

    # ---- barcode ---------------------------------------------------------
    def normalize_signal(self, sig):
        sig = np.ascontiguousarray(sig, dtype=np.float32)
        out = np.zeros_like(sig)
        self.L.pxo_normalize_signal(_p(sig), len(sig), _p(out))
        return out

    def barcode_window(self, adapter_signal):
        sig = np.ascontiguousarray(adapter_signal, dtype=np.float32)
        out = np.zeros(self.cfg.signal_trim_length, dtype=np.float32)
        pushed = self.L.pxo_barcode_window(C.byref(self.cfg), _p(sig), len(sig), _p(out))
        return out, bool(pushed)

    def phred(self, score):
        return self.L.pxo_phred(C.byref(self.cfg), float(np.float32(score)))

    # ---- events / polyA --------------------------------------------------
    def medfilt(self, x, k):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros_like(x)
        self.L.pxo_medfilt(_p(x), len(x), k, _p(out))
        return out

    def detect_events(self, sig, w1=None, w2=None, t1=None, t2=None, ph=None):
        c = self.cfg
        sig = np.ascontiguousarray(sig, dtype=np.float32)
        out = np.zeros(len(sig) + 1, dtype=N.EVENT_DTYPE)
        n = self.L.pxo_detect_events(
            _p(sig), len(sig), w1 or c.ed_window_length1, w2 or c.ed_window_length2,
            c.ed_threshold1 if t1 is None else t1, c.ed_threshold2 if t2 is None else t2,
            c.ed_peak_height if ph is None else ph, _p(out), len(out))
        return out[:n]

    def best_polya_interval(self, is_polya, length):
        ip = np.ascontiguousarray(is_polya, dtype=np.uint8)
        ln = np.ascontiguousarray(length, dtype=np.float32)
        i, j = C.c_int(0), C.c_int(0)
        ok = self.L.pxo_best_polya_interval(C.byref(self.cfg), _p(ip), _p(ln), len(ip),
                                            C.byref(i), C.byref(j))
        return (i.value, j.value) if ok else None

    def polya(self, scaled_full, rough_begin, rough_end, sampling_rate):
        x = np.ascontiguousarray(scaled_full, dtype=np.float32)
        r = np.zeros(1, dtype=N.RESULT_DTYPE)
        cap = SPIKE_ROWS
        while True:                     # every spike is counted; rerun with room for all of them
            sp = np.zeros((cap, 4), dtype=np.float32)
            self.L.pxo_polya(C.byref(self.cfg), _p(x), len(x), int(rough_begin),
                             -1 if rough_end is None else int(rough_end),
                             float(sampling_rate), _p(r), _p(sp), cap)
            if r[0]['polya_n_spikes'] <= cap:
                return r[0], sp
            cap = int(r[0]['polya_n_spikes'])

    # ---- events table / pseudo-fusion ------------------------------------
    def guppy_event_means(self, raw, calib_row, first_sample, n_events, scale, shift, stride=15):
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        mean = np.zeros(n_events, dtype=np.float32)
        scaled = np.zeros(n_events, dtype=np.float32)
        rc = self.L.pxo_guppy_event_means(_p(raw), len(raw), _p(_cal(calib_row)), int(first_sample),
                                          int(n_events), stride, np.float32(scale),
                                          np.float32(shift), _p(mean), _p(scaled))
        if rc != 0:
            raise Exception('Numbers of events and raw data strides does not match.')
        return mean, scaled

    def guppy_event_table(self, raw, calib_row, first_sample, n_events, scale, shift, stride=15):
        """(mean, stdv, scaled_mean) of convert_events_guppy + load_events."""
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        mean, stdv, scaled = (np.zeros(n_events, dtype=np.float32) for _ in range(3))
        rc = self.L.pxo_guppy_event_table(_p(raw), len(raw), _p(_cal(calib_row)), int(first_sample),
                                          int(n_events), stride, np.float32(scale),
                                          np.float32(shift), _p(mean), _p(stdv), _p(scaled))
        if rc != 0:
            raise Exception('Numbers of events and raw data strides does not match.')
        return mean, stdv, scaled

    def unsplit_scan(self, scaled_mean, first_sample, payload_start, sampling_rate, stride=15):
        sm = np.ascontiguousarray(scaled_mean, dtype=np.float32)
        cap = 4096                    # more in-read adapters than any read can hold
        iv = np.zeros((cap, 2), dtype=np.int64)
        n = self.L.pxo_unsplit_scan(C.byref(self.cfg), _p(sm), len(sm), int(first_sample), stride,
                                    int(payload_start), float(sampling_rate), _p(iv), cap)
        assert n <= cap
        return iv[:n], n

    def unsplit_scan_events(self, mean, starts, scale, shift, payload_start, sampling_rate):
        """a19 over a table with its own events (albacore): float32 `mean', ascending int64 `start'."""
        mean = np.ascontiguousarray(mean, dtype=np.float32)
        sm = np.float32(scale) * mean                      # float32: fl(fl(scale * mean) + shift)
        sm = sm + np.float32(shift)
        st = np.ascontiguousarray(starts, dtype=np.int64)
        cap = 4096
        iv = np.zeros((cap, 2), dtype=np.int64)
        n = self.L.pxo_unsplit_scan_events(C.byref(self.cfg), _p(sm), _p(st), len(sm), int(payload_start),
                                           float(sampling_rate), _p(iv), cap)
        assert n <= cap
        return iv[:n], n

    # ---- whole path ------------------------------------------------------
    def process_batch(self, arena, offsets, calib, scale_shift=None,
                      stage_mask=N.STAGE_ALL_DEMUX, want_spikes=False):
        arena = np.ascontiguousarray(arena, dtype=np.int16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        calib = np.ascontiguousarray(calib, dtype=N.CALIB_DTYPE)
        n = len(offsets) - 1
        if scale_shift is not None:
            scale_shift = np.ascontiguousarray(scale_shift, dtype=np.float32).reshape(n, 2)
        out = np.zeros(n, dtype=N.RESULT_DTYPE)
        cap = SPIKE_ROWS
        while True:
            spikes = np.zeros((n, cap, 4), dtype=np.float32) if want_spikes else None
            self.L.pxo_process_batch(C.byref(self.cfg), n, _p(arena), _p(offsets), _p(calib),
                                     _p(scale_shift), stage_mask, _p(out), _p(spikes), cap)
            most = int(out['polya_n_spikes'].max()) if n else 0
            if not want_spikes or most <= cap:
                break
            cap = most                  # a tail with more spikes than the rows on offer: all of them
        return (out, spikes) if want_spikes else out


def spikes_csr(records, spikes):
    """The oracle's dense [n, cap, 4] spike table as the product returns it: (rows [total, 4]
    in read order, offsets [n + 1])."""
    ns = np.where(records['polya_called'] != 0, records['polya_n_spikes'], 0).astype(np.int64)
    off = np.zeros(len(ns) + 1, dtype=np.int64)
    np.cumsum(ns, out=off[1:])
    rows = np.concatenate([spikes[i, :ns[i]] for i in range(len(ns))]) if len(ns) and off[-1] \
        else np.zeros((0, 4), dtype=np.float32)
    return rows.reshape(-1, 4), off


def reference_detect_events(sig, w1=7, w2=20, t1=3.0, t2=8.0, ph=4.0):
    """Call the REFERENCE's compiled detect_events (oracle/_ref), structs by
    value as in src/contrib/scrappie/{scrappie_structures,event_detection}.h."""
    if not os.path.isfile(REF_LIB_PATH):
        raise FileNotFoundError(REF_LIB_PATH)

    class RawTable(C.Structure):
        _fields_ = [('n', C.c_size_t), ('start', C.c_size_t), ('end', C.c_size_t),
                    ('raw', C.POINTER(C.c_float))]

    class EventTable(C.Structure):
        _fields_ = [('n', C.c_size_t), ('start', C.c_size_t), ('end', C.c_size_t),
                    ('event', C.POINTER(N.PxgEvent))]

    class DetectorParam(C.Structure):
        _fields_ = [('window_length1', C.c_size_t), ('window_length2', C.c_size_t),
                    ('threshold1', C.c_float), ('threshold2', C.c_float),
                    ('peak_height', C.c_float)]

    ref = C.CDLL(REF_LIB_PATH)
    ref.detect_events.restype = EventTable
    ref.detect_events.argtypes = [RawTable, DetectorParam]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    sig = np.ascontiguousarray(sig, dtype=np.float32)
    rt = RawTable(len(sig), 0, len(sig), sig.ctypes.data_as(C.POINTER(C.c_float)))
    et = ref.detect_events(rt, DetectorParam(w1, w2, t1, t2, ph))
    out = np.zeros(et.n, dtype=N.EVENT_DTYPE)
    if et.n:
        C.memmove(out.ctypes.data, et.event, et.n * C.sizeof(N.PxgEvent))
    libc.free(C.cast(et.event, C.c_void_p))
    return out
