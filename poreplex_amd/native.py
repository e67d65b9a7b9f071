"""ctypes binding of the C ABI in include/pxg.h (libpxg.so).

This is the thin host layer north_star asks for: Python marshals arrays, every
numeric stage runs in HIP behind ``pxg_*``.  The library is built in-tree by
``__graft_entry__.build()`` (hipcc --offload-arch=gfx950); if it is missing or
no GPU is usable this module fails LOUDLY -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from . import config as _config

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'csrc', 'libpxg.so')

PXG_ABI_VERSION = 5
LSTM_ARITH = {'q8': 0, 'f32': 1}      # enum pxg_lstm_arith
PXG_E_NOMEM, PXG_E_UNSUPPORTED = -4, -6
PXG_MAX_STATES = 8
PXG_MAX_MIXTURE = 4
PXG_N_SEGMENTS = 8
PXG_MAX_CLASSES = 8
PXG_MAX_CALIBRATION = 64
UNSPLIT_E_GEOMETRY = -3
UNSPLIT_E_WINDOW = -4

STATUS_NAMES = (
    'okay', 'disappeared', 'irregular_fast5', 'scaler_signal_too_short',
    'scaling_qc_fail', 'adapter_not_detected', 'not_basecalled',
    'basecall_table_incomplete', 'unsplit_read', 'sequence_too_short',
    'unknown_error')
STATUS_CODE = {name: i for i, name in enumerate(STATUS_NAMES)}

STAGE_SCALER = 1
STAGE_SEGMENT = 2
STAGE_BARCODE = 4
STAGE_POLYA = 8
STAGE_ALL_DEMUX = 7

TIMER_NAMES = ('head_pool', 'scaler_lstm', 'segment', 'barcode_window',
               'demux_bidir', 'demux_top', 'polya', 'event_means', 'unsplit',
               'finalize', 'total')


class PxgError(RuntimeError):
    pass


class PxgCalib(C.Structure):
    _fields_ = [('range', C.c_double), ('digitisation', C.c_double),
                ('offset', C.c_double), ('sampling_rate', C.c_double)]


class PxgHmm(C.Structure):
    _fields_ = [
        ('n_states', C.c_int32), ('adapter_state', C.c_int32),
        ('polya_state', C.c_int32), ('leader_low_state', C.c_int32),
        ('leader_high_state', C.c_int32), ('reserved', C.c_int32),
        ('name_rank', C.c_int32 * PXG_MAX_STATES),
        ('n_mix', C.c_int32 * PXG_MAX_STATES),
        ('start_prob', C.c_double * PXG_MAX_STATES),
        ('mix_mu', (C.c_double * PXG_MAX_MIXTURE) * PXG_MAX_STATES),
        ('mix_sigma', (C.c_double * PXG_MAX_MIXTURE) * PXG_MAX_STATES),
        ('mix_weight', (C.c_double * PXG_MAX_MIXTURE) * PXG_MAX_STATES),
        ('trans', (C.c_double * PXG_MAX_STATES) * PXG_MAX_STATES),
    ]


class PxgLstmLayer(C.Structure):
    _fields_ = [('input_dim', C.c_int32), ('units', C.c_int32),
                ('kernel', C.POINTER(C.c_float)), ('recurrent', C.POINTER(C.c_float)),
                ('bias', C.POINTER(C.c_float))]


class PxgDenseLayer(C.Structure):
    _fields_ = [('in_dim', C.c_int32), ('out_dim', C.c_int32),
                ('kernel', C.POINTER(C.c_float)), ('bias', C.POINTER(C.c_float))]


class PxgConfig(C.Structure):
    _fields_ = [
        ('abi_version', C.c_uint32), ('device_id', C.c_int32),
        ('stride', C.c_int32), ('scaler_length', C.c_int32),
        ('scaler_min_length', C.c_int32), ('lstm_arith', C.c_int32),
        ('scaler_xfrm', C.c_double * 4),
        ('scaler_qc_scale', C.c_double * 2), ('scaler_qc_shift', C.c_double * 2),
        ('scaler_lstm1', PxgLstmLayer), ('scaler_lstm2', PxgLstmLayer),
        ('scaler_dense', PxgDenseLayer),
        ('segmentation_scan_limit', C.c_int32), ('reserved1', C.c_int32),
        ('segmentation_model', PxgHmm), ('unsplit_model', PxgHmm),
        ('unsplit_window_size', C.c_double), ('unsplit_window_step', C.c_double),
        ('unsplit_strict_duration', C.c_double), ('unsplit_strict_full_length', C.c_double),
        ('unsplit_strict_dna_length', C.c_double), ('unsplit_loosen_full_length', C.c_double),
        ('unsplit_loosen_dna_length', C.c_double),
        ('number_of_decoy_labels', C.c_int32), ('number_of_barcodes', C.c_int32),
        ('minimum_dna_length', C.c_int32), ('maximum_dna_length', C.c_int32),
        ('signal_trim_length', C.c_int32), ('n_calibration', C.c_int32),
        ('calibration', C.c_double * PXG_MAX_CALIBRATION),
        ('score_threshold', C.c_double),
        ('pad_filler', C.c_float), ('reserved2', C.c_int32),
        ('demux_fwd', PxgLstmLayer), ('demux_bwd', PxgLstmLayer),
        ('demux_top', PxgLstmLayer), ('demux_dense', PxgDenseLayer),
        ('polya_refinement_expansion', C.c_int32), ('polya_openend_expansion', C.c_int32),
        ('polya_median_pre_filter', C.c_int32),
        ('polya_maximum_openend_extension', C.c_int32),
        ('ed_window_length1', C.c_int32), ('ed_window_length2', C.c_int32),
        ('ed_threshold1', C.c_float), ('ed_threshold2', C.c_float),
        ('ed_peak_height', C.c_float), ('polya_spike_tolerance', C.c_int32),
        ('polya_mean_dist', C.c_double * 2), ('polya_mean_z_cutoff', C.c_double),
        ('polya_stdv_max', C.c_double), ('polya_stdv_range', C.c_double * 2),
        ('polya_spike_weight', C.c_double),
        ('polya_mean_trigger_recalibration', C.c_double),
        ('recal_max_dist_from_adapter', C.c_int32), ('recal_min_length', C.c_int32),
        ('recal_max_stdv', C.c_double),
    ]


class PxgReadResult(C.Structure):
    _fields_ = [
        ('status', C.c_int32), ('n_pooled', C.c_int32),
        ('seg_first', C.c_int32 * PXG_N_SEGMENTS), ('seg_last', C.c_int32 * PXG_N_SEGMENTS),
        ('scale', C.c_float), ('shift', C.c_float), ('scaler_pred', C.c_float * 2),
        ('bc_pushed', C.c_int8), ('bc_called', C.c_int8), ('bc_label', C.c_int8),
        ('bc_phred', C.c_uint8), ('bc_score', C.c_float),
        ('probs', C.c_float * PXG_MAX_CLASSES),
        ('polya_called', C.c_int8), ('reserved8', C.c_int8), ('reserved16', C.c_int16),
        ('polya_n_spikes', C.c_int32),
        ('polya_dwell_samples', C.c_int32), ('reserved32', C.c_int32),
        ('polya_begin', C.c_int64), ('polya_end', C.c_int64),
    ]


class PxgBatchExtras(C.Structure):          # pxg_batch_extras (pxg_process_batch_ex)
    _fields_ = [('struct_bytes', C.c_uint32), ('unsplit_block_stride', C.c_int32),
                ('scale_shift_or_null', C.c_void_p),
                ('z', C.c_void_p), ('z_bytes', C.c_int64), ('chunks', C.c_void_p), ('n_chunks', C.c_int64),
                ('data_base', C.c_int64), ('dst_base', C.c_int64),
                ('unsplit_first_sample', C.c_void_p), ('unsplit_n_blocks', C.c_void_p),
                ('unsplit_cap', C.c_int64), ('unsplit_intervals', C.c_void_p),
                ('unsplit_count', C.c_void_p), ('unsplit_total', C.c_int64),
                ('spike_cap', C.c_int64), ('spikes', C.c_void_p), ('spike_offsets', C.c_void_p),
                ('spike_total', C.c_int64)]


class PxgEvent(C.Structure):
    _fields_ = [('start', C.c_uint64), ('length', C.c_float), ('mean', C.c_float),
                ('stdv', C.c_float), ('pos', C.c_int32), ('state', C.c_int32)]


class PxgTextColumn(C.Structure):          # one NumPy '<U' array: UCS-4, fixed width
    _fields_ = [('data', C.c_void_p), ('width', C.c_int64)]


class PxgSummaryColumns(C.Structure):       # pxg_summary_columns
    _fields_ = [('n', C.c_int64), ('string_row', C.c_void_p), ('text', PxgTextColumn * 5),
                ('start_time', C.c_void_p), ('sampling_rate', C.c_void_p), ('duration', C.c_void_p),
                ('has_summary', C.c_void_p), ('num_events', C.c_void_p), ('sequence_length', C.c_void_p),
                ('mean_qscore', C.c_void_p),
                ('status', C.c_void_p), ('status_names', C.POINTER(C.c_char_p)), ('n_status', C.c_int32),
                ('label', C.c_void_p), ('label_names', C.POINTER(C.c_char_p)), ('n_labels', C.c_int32),
                ('barcode', C.c_void_p), ('barcode_score', C.c_void_p),
                ('barcode_names', C.POINTER(C.c_char_p)), ('n_barcode_names', C.c_int32),
                ('has_polya', C.c_void_p), ('polya_dwell', C.c_void_p)]


class PxgStageTimes(C.Structure):
    _fields_ = [('ms', C.c_float * len(TIMER_NAMES)),
                ('n_launches', C.c_int64 * len(TIMER_NAMES))]


class PxgDeviceInfo(C.Structure):
    _fields_ = [('name', C.c_char * 128), ('arch', C.c_char * 32),
                ('compute_units', C.c_int32), ('wavefront_size', C.c_int32),
                ('total_mem', C.c_int64), ('lds_per_cu', C.c_int32),
                ('clock_khz', C.c_int32)]


# numpy views of the POD records (same layout, checked against ctypes sizes)
RESULT_DTYPE = np.dtype([
    ('status', '<i4'), ('n_pooled', '<i4'),
    ('seg_first', '<i4', (PXG_N_SEGMENTS,)), ('seg_last', '<i4', (PXG_N_SEGMENTS,)),
    ('scale', '<f4'), ('shift', '<f4'), ('scaler_pred', '<f4', (2,)),
    ('bc_pushed', 'i1'), ('bc_called', 'i1'), ('bc_label', 'i1'), ('bc_phred', 'u1'),
    ('bc_score', '<f4'), ('probs', '<f4', (PXG_MAX_CLASSES,)),
    ('polya_called', 'i1'), ('reserved8', 'i1'), ('reserved16', '<i2'), ('polya_n_spikes', '<i4'),
    ('polya_dwell_samples', '<i4'), ('reserved32', '<i4'), ('polya_begin', '<i8'), ('polya_end', '<i8'),
], align=True)
CALIB_DTYPE = np.dtype([('range', '<f8'), ('digitisation', '<f8'),
                        ('offset', '<f8'), ('sampling_rate', '<f8')])
EVENT_DTYPE = np.dtype([('start', '<u8'), ('length', '<f4'), ('mean', '<f4'),
                        ('stdv', '<f4'), ('pos', '<i4'), ('state', '<i4')], align=True)
assert RESULT_DTYPE.itemsize == C.sizeof(PxgReadResult), \
    (RESULT_DTYPE.itemsize, C.sizeof(PxgReadResult))
assert EVENT_DTYPE.itemsize == C.sizeof(PxgEvent)
assert CALIB_DTYPE.itemsize == C.sizeof(PxgCalib)


# --------------------------------------------------------------------------
# config dict -> pxg_config
# --------------------------------------------------------------------------
def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _fill_hmm(hmm, modeldata):
    """worker_persistence.py:95-121 load_segmentation_model, as tables."""
    names = [s['name'] for s in modeldata]
    if len(names) > PXG_MAX_STATES:
        raise ValueError('HMM has more than {} states'.format(PXG_MAX_STATES))
    index = {n: i for i, n in enumerate(names)}
    hmm.n_states = len(names)
    hmm.adapter_state = index.get('adapter', -1)
    hmm.polya_state = index.get('polya-tail', -1)
    hmm.leader_low_state = index.get('leader-low', -1)
    hmm.leader_high_state = index.get('leader-high', -1)
    for rank, name in enumerate(sorted(names)):
        hmm.name_rank[index[name]] = rank
    for i, s in enumerate(modeldata):
        em = s['emission']
        if len(em) > PXG_MAX_MIXTURE:
            raise ValueError('too many mixture components')
        hmm.n_mix[i] = len(em)
        for k, row in enumerate(em):
            hmm.mix_mu[i][k] = float(row[0])
            hmm.mix_sigma[i][k] = float(row[1])
            hmm.mix_weight[i][k] = float(row[2]) if len(row) > 2 else 1.0
        hmm.start_prob[i] = float(s.get('start_prob', 0.0))
        for nextstate, prob in s['transition']:
            hmm.trans[i][index[nextstate]] = float(prob)
    return names


def _lstm(layer, kernel, recurrent, bias, keep):
    kernel, recurrent, bias = _f32(kernel), _f32(recurrent), _f32(bias)
    keep += [kernel, recurrent, bias]
    layer.input_dim = kernel.shape[0]
    layer.units = recurrent.shape[0]
    assert kernel.shape[1] == 4 * layer.units == recurrent.shape[1] == bias.shape[0]
    layer.kernel, layer.recurrent, layer.bias = _fptr(kernel), _fptr(recurrent), _fptr(bias)


def _dense(layer, kernel, bias, keep):
    kernel, bias = _f32(kernel), _f32(bias)
    keep += [kernel, bias]
    layer.in_dim, layer.out_dim = kernel.shape
    layer.kernel, layer.bias = _fptr(kernel), _fptr(bias)


class NativeConfig:
    """pxg_config plus the NumPy arrays its pointers reference."""

    def __init__(self, config, device_id=0):
        self.keep = []
        self.struct = cfg = PxgConfig()
        cfg.abi_version = PXG_ABI_VERSION
        cfg.device_id = device_id

        sp = config['signal_processing']
        # arithmetic of the recurrent matmuls (include/pxg.h pxg_lstm_arith): 'q8' = exact fixed point on
        # the int8 matrix pipe (default), 'f32' = float32 fma chains; PXG_LSTM_ARITH overrides the config
        arith = os.environ.get('PXG_LSTM_ARITH') or str(sp.get('lstm_arith', 'q8'))
        if arith not in LSTM_ARITH:
            raise ValueError('lstm_arith must be one of %s' % sorted(LSTM_ARITH))
        cfg.lstm_arith = LSTM_ARITH[arith]
        scaler = _config.load_model_arrays(sp['scaler_model'])
        cfg.stride = int(sp['rough_signal_stride'])
        cfg.scaler_length = int(scaler['input_length'])
        cfg.scaler_min_length = int(scaler['input_min_length'])
        if int(scaler['input_stride']) != cfg.stride:
            raise ValueError('scaler stride differs from rough_signal_stride')
        xfrm = [float(v) for v in scaler['output_transform']]
        for i, v in enumerate(xfrm):
            cfg.scaler_xfrm[i] = v
        q = float(sp['scaler_qc_threshold'])          # signal_loader.py:65-68
        for i, qq in enumerate((q, 1 - q)):
            cfg.scaler_qc_scale[i] = _config.norm_ppf(qq, xfrm[0], xfrm[1])
            cfg.scaler_qc_shift[i] = _config.norm_ppf(qq, xfrm[2], xfrm[3])
        _lstm(cfg.scaler_lstm1, scaler['lstm1_kernel'], scaler['lstm1_recurrent'],
              scaler['lstm1_bias'], self.keep)
        _lstm(cfg.scaler_lstm2, scaler['lstm2_kernel'], scaler['lstm2_recurrent'],
              scaler['lstm2_bias'], self.keep)
        _dense(cfg.scaler_dense, scaler['dense_kernel'], scaler['dense_bias'], self.keep)

        cfg.segmentation_scan_limit = int(config['segmentation']['segmentation_scan_limit'])
        self.state_names = _fill_hmm(cfg.segmentation_model, config['segmentation_model'])
        self.unsplit_state_names = _fill_hmm(cfg.unsplit_model,
                                             config['unsplit_read_detection_model'])

        ur = config['unsplit_read_detection']        # signal_analyzer.py:374-383
        for key in ('window_size', 'window_step', 'strict_duration', 'strict_full_length',
                    'strict_dna_length', 'loosen_full_length', 'loosen_dna_length'):
            setattr(cfg, 'unsplit_' + key, float(ur[key]))

        dm = config['demultiplexing']
        demux = _config.load_model_arrays(dm['demux_model'])
        cfg.number_of_decoy_labels = int(dm['number_of_decoy_labels'])
        cfg.number_of_barcodes = int(dm['number_of_barcodes'])
        cfg.minimum_dna_length = int(dm['minimum_dna_length'])
        cfg.maximum_dna_length = int(dm['maximum_dna_length'])
        cfg.signal_trim_length = int(dm['signal_trim_length'])
        calib = np.asarray(demux['calibration'], dtype=np.float64)
        if len(calib) > PXG_MAX_CALIBRATION:
            raise ValueError('calibration table too long')
        cfg.n_calibration = len(calib)
        for i, v in enumerate(calib):
            cfg.calibration[i] = float(v)
        qfilter = int(config.get('barcoding_quality_filter', 18))
        if len(calib) - 1 < qfilter:                  # barcoding.py:41-45
            raise ValueError('The current demultiplexer does not support calibrated score '
                             'of {}. Consider lowering --barcoding-quality-filter value.'
                             .format(qfilter))
        cfg.score_threshold = float(calib[qfilter])
        cfg.pad_filler = -1000.0                      # barcoding.py:32
        _lstm(cfg.demux_fwd, demux['fwd_kernel'], demux['fwd_recurrent'],
              demux['fwd_bias'], self.keep)
        _lstm(cfg.demux_bwd, demux['bwd_kernel'], demux['bwd_recurrent'],
              demux['bwd_bias'], self.keep)
        _lstm(cfg.demux_top, demux['top_kernel'], demux['top_recurrent'],
              demux['top_bias'], self.keep)
        _dense(cfg.demux_dense, demux['dense_kernel'], demux['dense_bias'], self.keep)
        self.calibration = calib

        pa = config['polya_dwell']                    # polya.py:32-48
        cfg.polya_refinement_expansion = int(pa['refinement_expansion'])
        cfg.polya_openend_expansion = int(pa['openend_expansion'])
        cfg.polya_median_pre_filter = int(pa['median_pre_filter'])
        cfg.polya_maximum_openend_extension = int(pa['maximum_openend_extension'])
        ed = pa['event_detection']
        cfg.ed_window_length1 = int(ed['window_length1'])
        cfg.ed_window_length2 = int(ed['window_length2'])
        cfg.ed_threshold1 = float(ed['threshold1'])
        cfg.ed_threshold2 = float(ed['threshold2'])
        cfg.ed_peak_height = float(ed['peak_height'])
        cfg.polya_spike_tolerance = int(pa['spike_tolerance'])
        cfg.polya_mean_dist[0], cfg.polya_mean_dist[1] = map(float, pa['polya_mean_dist'])
        cfg.polya_mean_z_cutoff = float(pa['polya_mean_z_cutoff'])
        cfg.polya_stdv_max = float(pa['polya_stdv_max'])
        cfg.polya_stdv_range[0], cfg.polya_stdv_range[1] = map(float, pa['polya_stdv_range'])
        cfg.polya_spike_weight = float(pa['spike_weight'])
        cfg.polya_mean_trigger_recalibration = float(pa['polya_mean_trigger_recalibration'])
        rc = pa['recalibrate_shifted_signal']
        cfg.recal_max_dist_from_adapter = int(rc['max_dist_from_adapter'])
        cfg.recal_min_length = int(rc['min_length'])
        cfg.recal_max_stdv = float(rc['max_stdv'])


# --------------------------------------------------------------------------
# library loading
# --------------------------------------------------------------------------
_lib = None

TEXT_LIB_PATH = os.path.join(HERE, 'csrc', 'libpxghost.so')
_TEXT_SIGNATURES = {      # libpxghost.so: host-only helpers (sink text, sample codec)
    'pxg_summary_rows': (C.c_int64, [C.POINTER(PxgSummaryColumns), C.c_char_p, C.c_int64]),
    'pxg_z_count_chunks': (C.c_int64, [C.c_int64, C.c_void_p]),
    'pxg_z_encode': (C.c_int64, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'pxg_z_encode_as': (C.c_int64, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]),
    'pxg_z_decode': (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    'pxg_z_decode_n': (C.c_int, [C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    'pxg_z_validate': (C.c_int, [C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    'pxg_h5_open': (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    'pxg_h5_open_mt': (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]),
    'pxg_h5_close': (None, [C.c_void_p]),
    'pxg_h5_last_error': (C.c_char_p, []),
    'pxg_h5_n_reads': (C.c_int64, [C.c_void_p]),
    'pxg_h5_is_multi': (C.c_int, [C.c_void_p]),
    'pxg_h5_read_id': (C.c_int, [C.c_void_p, C.c_int64, C.c_char_p, C.c_int64]),
    'pxg_h5_read_ids': (C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
    'pxg_h5_info': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    'pxg_h5_info_mt': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32]),
    'pxg_h5_open_many': (C.c_int, [C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    'pxg_h5_close_many': (None, [C.c_int64, C.c_void_p]),
    'pxg_h5_basecall': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.POINTER(C.c_int32)]),
    'pxg_h5_events': (C.c_int64, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    'pxg_h5_load_signals': (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int32, C.c_void_p]),
    'pxg_h5_basecall_many': (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
}
_text_lib = None


class PxgH5ReadInfo(C.Structure):           # pxg_h5_read_info
    _fields_ = [('status', C.c_int32), ('bc_present', C.c_int32), ('read_id', C.c_char * 64),
                ('channel_number', C.c_char * 16), ('run_id', C.c_char * 64), ('sample_id', C.c_char * 128),
                ('error', C.c_char * 160), ('duration', C.c_int64), ('start_time', C.c_int64),
                ('n_samples', C.c_int64), ('calib', PxgCalib), ('bc_table', C.c_int32),
                ('bc_block_stride', C.c_int32), ('bc_sequence_length', C.c_int64), ('bc_num_events', C.c_int64),
                ('bc_first_sample', C.c_int64), ('bc_n_moves', C.c_int64), ('bc_move_sum', C.c_int64),
                ('bc_seq_len', C.c_int64), ('bc_mean_qscore', C.c_double)]


H5_INFO_DTYPE = np.dtype([
    ('status', '<i4'), ('bc_present', '<i4'), ('read_id', 'S64'), ('channel_number', 'S16'), ('run_id', 'S64'),
    ('sample_id', 'S128'), ('error', 'S160'), ('duration', '<i8'), ('start_time', '<i8'), ('n_samples', '<i8'),
    ('calib', [('range', '<f8'), ('digitisation', '<f8'), ('offset', '<f8'), ('sampling_rate', '<f8')]),
    ('bc_table', '<i4'), ('bc_block_stride', '<i4'), ('bc_sequence_length', '<i8'), ('bc_num_events', '<i8'),
    ('bc_first_sample', '<i8'), ('bc_n_moves', '<i8'), ('bc_move_sum', '<i8'), ('bc_seq_len', '<i8'),
    ('bc_mean_qscore', '<f8')], align=True)
assert H5_INFO_DTYPE.itemsize == C.sizeof(PxgH5ReadInfo), (H5_INFO_DTYPE.itemsize, C.sizeof(PxgH5ReadInfo))

_SIGNATURES = {
    'pxg_create': (C.c_int, [C.POINTER(PxgConfig), C.POINTER(C.c_void_p)]),
    'pxg_destroy': (None, [C.c_void_p]),
    'pxg_last_error': (C.c_char_p, [C.c_void_p]),
    'pxg_abi_version': (C.c_int, []),
    'pxg_get_device_info': (C.c_int, [C.c_void_p, C.POINTER(PxgDeviceInfo)]),
    'pxg_device_pci_bus_id': (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    'pxg_process_batch': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    'pxg_batch_upload': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    'pxg_batch_stage': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    'pxg_batch_stage_prefix': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int64]),
    'pxg_batch_stage_z_prefix': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                           C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    'pxg_batch_download_samples': (C.c_int, [C.c_void_p, C.c_void_p]),
    'pxg_batch_stage_z': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                    C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pxg_batch_upload_tiled': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    'pxg_batch_swap': (C.c_int, [C.c_void_p]),
    'pxg_host_register': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    'pxg_host_unregister': (C.c_int, [C.c_void_p, C.c_void_p]),
    'pxg_batch_run': (C.c_int, [C.c_void_p, C.c_uint32]),
    'pxg_batch_sync': (C.c_int, [C.c_void_p]),
    'pxg_batch_download': (C.c_int, [C.c_void_p, C.c_void_p]),
    'pxg_batch_download_spikes': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'pxg_process_batch_ex': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                       C.POINTER(PxgBatchExtras), C.c_void_p]),
    'pxg_merge_stats': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'pxg_batch_times': (C.c_int, [C.c_void_p, C.POINTER(PxgStageTimes)]),
    'pxg_raw_to_pa': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pxg_head_pool': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    'pxg_scaler_lstm': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'pxg_scaler_transform': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    'pxg_pool_scale': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    'pxg_viterbi': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pxg_barcode_window': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    'pxg_demux_lstm': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'pxg_guppy_event_means': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                        C.c_void_p]),
    'pxg_batch_unsplit_scan': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                         C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    'pxg_batch_unsplit_scan_events': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    'pxg_ctx_lock': (C.c_int, [C.c_void_p, C.c_int]),
    'pxg_ctx_unlock': (C.c_int, [C.c_void_p, C.c_int]),
    'pxg_batch_pooled_signal': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pxg_batch_download_windows': (C.c_int, [C.c_void_p, C.c_void_p]),
    'pxg_batch_event_table': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    'pxg_polya': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'pxg_detect_events': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_int64, C.c_void_p, C.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)
TEXT_SYMBOLS = tuple(_TEXT_SIGNATURES)


def load_library(path=None):
    """Load libpxg.so; raise PxgError (never fall back) when it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get('PXG_LIBRARY') or LIB_PATH   # A/B builds for profiling
    if not os.path.isfile(path):
        raise PxgError(
            'HIP extension {} is missing: run `python -c "import __graft_entry__ as g; '
            'g.build()"` (hipcc --offload-arch=gfx950). There is no CPU fallback.'.format(path))
    lib = C.CDLL(path)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError => ABI mismatch, also loud
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.pxg_abi_version() != PXG_ABI_VERSION:
        raise PxgError('libpxg.so ABI {} != binding ABI {}'.format(
            lib.pxg_abi_version(), PXG_ABI_VERSION))
    _lib = lib
    return lib


def device_pci_bus_id(device=0):
    """PCI address of HIP device `device` ('0000:c1:00.0'), or None when it cannot be told."""
    buf = C.create_string_buffer(64)
    try:
        if load_library().pxg_device_pci_bus_id(int(device), buf, 64) != 0:
            return None
    except PxgError:
        return None
    return buf.value.decode().lower() or None


def load_text_library(path=None):
    """Load libpxghost.so (built next to libpxg.so by the same Makefile); loud when absent."""
    global _text_lib
    if _text_lib is not None and path is None:
        return _text_lib
    path = path or TEXT_LIB_PATH
    if not os.path.isfile(path):
        raise PxgError('{} is missing: run `python -c "import __graft_entry__ as g; g.build()"`'.format(path))
    lib = C.CDLL(path)
    for name, (restype, argtypes) in _TEXT_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _text_lib = lib
    return lib


_pyhost = False


def load_pyhost():
    """The CPython extension csrc/_pxgpy (result dicts from columns, pxg_pyreport.c) for THIS
    interpreter, or None: it is a host-side accelerator with a Python fallback, not part of the
    numeric path (PXG_NO_PYHOST=1 forces the fallback, for the equivalence tests)."""
    global _pyhost
    if _pyhost is False:
        _pyhost = None
        import importlib.machinery
        import importlib.util
        if not os.environ.get('PXG_NO_PYHOST'):
            for suffix in importlib.machinery.EXTENSION_SUFFIXES:
                path = os.path.join(HERE, 'csrc', '_pxgpy' + suffix)
                if os.path.isfile(path):
                    try:
                        spec = importlib.util.spec_from_file_location('_pxgpy', path)
                        mod = importlib.util.module_from_spec(spec)
                        spec.loader.exec_module(mod)
                        _pyhost = mod
                    except ImportError:
                        pass
                    break
    return _pyhost


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def page_exclusive(n, dtype, fill=None):
    """An array of n items on pages of its own, from an ANONYMOUS MAPPING -- what a caller hands to
    NativeContext.pin().  Never from the malloc heap: hipHostRegister works on pages (a page-locked page that also holds
    somebody else's array makes the runtime treat that array as page-locked too), and a brk-heap range that was
    registered and unregistered once must not be page-locked again -- ROCm 7.2 locks a big pageable source of a copy
    in place, and when that lock lands on such pages the DMA engine faults inside the range (profiles/r05/fault_hunt.md).
    A mapping goes back to the kernel with the array; the heap's pages stay what they are."""
    import mmap
    dtype = np.dtype(dtype)
    nbytes = int(n) * dtype.itemsize
    m = mmap.mmap(-1, max(nbytes, 1), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)      # (page-aligned, whole pages)
    if nbytes >= (8 << 20) and hasattr(mmap, 'MADV_HUGEPAGE') and not os.environ.get('PXG_NO_HUGE_PAGES'):
        try:                    # a staging arena: 650 first touches of 2 MB instead of 330 000 of 4 KB
            m.madvise(mmap.MADV_HUGEPAGE)
        except OSError:
            pass
    out = np.frombuffer(m, dtype=np.uint8)[:nbytes].view(dtype)
    if fill is not None:
        out[...] = fill
    return out          # (the .base chain keeps the mapping alive)


def pinnable(array):
    """`array` itself when it is large enough to have been mapped on its own (glibc maps allocations above 32 MB
    whatever its dynamic threshold says), otherwise a page_exclusive() copy of it: what to page-lock instead of an
    array that may live in the malloc heap."""
    if array.nbytes >= (64 << 20) or array.nbytes == 0:
        return array
    out = page_exclusive(array.size, array.dtype).reshape(array.shape)
    out[...] = array
    return out


# ---- compressed samples (include/pxg.h, pxg_zcodec.cpp) ---------------------------------------
Z_CHUNK = 1024
Z_CHUNK_DTYPE = np.dtype([('data_off', np.int64), ('dst', np.int64), ('first', np.int16),
                          ('len', np.int16), ('codec', np.int32)])
Z_CODECS = {'bytes': 0, 'packed': 1}      # PXG_Z_BYTES (one or two bytes per delta), PXG_Z_PACKED (bit widths per 4 deltas)


def z_encode(arena, offsets, codec='packed'):
    """int16 samples of many reads -> (bytes uint8[], chunk records, chunk_base int64[n + 1]):
    chunk_base[r] = index of read r's first chunk.  Host side, offline (bundle writing).  `codec`: 'packed'
    (bit-packed deltas, ~0.96 bytes per sample) or 'bytes' (VBZ's byte codes, ~1.19; what bundles written
    before ABI 5 hold -- the decoders take either, chunk by chunk)."""
    lib = load_text_library()
    arena = np.ascontiguousarray(arena, dtype=np.int16)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    per_read = (np.diff(offsets) + Z_CHUNK - 1) // Z_CHUNK
    chunk_base = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(per_read, out=chunk_base[1:])
    n_chunks = int(chunk_base[-1])
    chunks = np.zeros(n_chunks, dtype=Z_CHUNK_DTYPE)
    out = np.empty(n_chunks * (Z_CHUNK // 8 + 8) + 2 * (len(arena) + 3 * n_chunks) + 2 * Z_CHUNK + 16, dtype=np.uint8)
    got = lib.pxg_z_encode_as(n, _ptr(arena), _ptr(offsets), _ptr(out), len(out), _ptr(chunks), Z_CODECS[codec])
    if got < 0:
        raise PxgError('pxg_z_encode_as failed ({})'.format(got))
    return out[:got].copy(), chunks, chunk_base


def z_validate(z, chunks, n_samples, data_base=0, dst_base=0):
    """Raise PxgError unless the chunk records tile [dst_base, dst_base + n_samples) and stay
    inside the bytes of `z` (records and bytes come from files: a truncated bundle must not
    become an out-of-bounds access in a decoder)."""
    lib = load_text_library()
    chunks = np.ascontiguousarray(chunks, dtype=Z_CHUNK_DTYPE)
    rc = lib.pxg_z_validate(len(chunks), _ptr(chunks), int(data_base), int(len(z)), int(dst_base),
                            int(n_samples))
    if rc:
        raise PxgError('encoded samples: the chunk records do not describe the byte stream '
                       '(truncated or corrupt bundle)')


def z_decode(z, chunks, n_samples, data_base=0, dst_base=0):
    """Reference decoder (host): the int16 samples the chunk records describe."""
    lib = load_text_library()
    z = np.ascontiguousarray(z, dtype=np.uint8)
    chunks = np.ascontiguousarray(chunks, dtype=Z_CHUNK_DTYPE)
    z_validate(z, chunks, n_samples, data_base, dst_base)
    out = np.zeros(int(n_samples), dtype=np.int16)
    rc = lib.pxg_z_decode_n(len(chunks), _ptr(z), len(z), _ptr(chunks), int(data_base), int(dst_base), _ptr(out))
    if rc:
        raise PxgError('pxg_z_decode failed ({})'.format(rc))
    return out


class EncodedSamples:
    """The samples of a run of consecutive reads as the encoded bytes + chunk records of a
    bundle (views, nothing copied): what ReadBundle hands to the loader instead of an int16
    arena, and NativeContext.stage_z sends across the link."""

    def __init__(self, z, chunks, data_base, dst_base, n_samples):
        self.z, self.chunks = z, chunks
        self.data_base, self.dst_base, self.n_samples = int(data_base), int(dst_base), int(n_samples)

    def __len__(self):
        return self.n_samples

    def decode(self):
        return z_decode(self.z, self.chunks, self.n_samples, self.data_base, self.dst_base)


def _names(strings):
    arr = (C.c_char_p * len(strings))(*[s.encode('ascii') for s in strings])
    return arr


def summary_rows(text_columns, string_row, start_time, sampling_rate, duration, has_summary,
                 num_events, sequence_length, mean_qscore, status, status_names, label, label_names,
                 barcode=None, barcode_score=None, barcode_names=None, has_polya=None, polya_dwell=None,
                 scratch=None):
    """pxg_summary_rows: the sequencing_summary.txt rows of n reads as one bytes object, or
    None when the library declines (non-ASCII text: the caller formats in Python).
    text_columns = the five NumPy '<U' arrays filename, read_id, run_id, channel, sample_id;
    string_row[k] = the element of those arrays row k prints.  `scratch`: a one-element list
    that keeps the output buffer between calls (a fresh 3 MB buffer per batch is page faults)."""
    lib = load_text_library()
    n = len(string_row)
    keep = []                                     # keeps the converted arrays alive over the call

    def col(a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)
    c = PxgSummaryColumns()
    c.n = n
    c.string_row = col(string_row, np.int64)
    width = 0
    for f, arr in enumerate(text_columns):
        if arr.dtype.kind != 'U' or arr.dtype.byteorder not in ('<', '='):
            return None
        arr = np.ascontiguousarray(arr)
        keep.append(arr)
        c.text[f].data = arr.ctypes.data_as(C.c_void_p)
        c.text[f].width = arr.dtype.itemsize // 4
        width += arr.dtype.itemsize // 4
    c.start_time, c.sampling_rate = col(start_time, np.int64), col(sampling_rate, np.float64)
    c.duration, c.has_summary = col(duration, np.int64), col(has_summary, np.uint8)
    c.num_events, c.sequence_length = col(num_events, np.int64), col(sequence_length, np.int64)
    c.mean_qscore = col(mean_qscore, np.float64)
    sn, ln = _names(status_names), _names(label_names)
    c.status, c.status_names, c.n_status = col(status, np.int32), sn, len(status_names)
    c.label, c.label_names, c.n_labels = col(label, np.int32), ln, len(label_names)
    extra = max(len(x) for x in status_names) + max(len(x) for x in label_names)
    if barcode_names is not None:
        bn = _names(barcode_names)
        c.barcode, c.barcode_score = col(barcode, np.int32), col(barcode_score, np.int32)
        c.barcode_names, c.n_barcode_names = bn, len(barcode_names)
        extra += max(len(x) for x in barcode_names) + 12
    if polya_dwell is not None:
        c.has_polya, c.polya_dwell = col(has_polya, np.uint8), col(polya_dwell, np.float64)
        extra += 32
    cap = n * (width + extra + 140) + 64          # 4 integers + 2 floats of <= 25 characters, separators
    if scratch is not None and scratch and len(scratch[0]) >= cap:
        buf = scratch[0]
    else:
        buf = C.create_string_buffer(cap + cap // 4)
        if scratch is not None:
            scratch[:] = [buf]
    got = lib.pxg_summary_rows(C.byref(c), buf, len(buf))
    if got == PXG_E_UNSUPPORTED:
        return None
    if got < 0:
        raise PxgError('pxg_summary_rows failed ({})'.format(got))
    return C.string_at(buf, got)


def pack_reads(signals):
    """list of int16 arrays -> (arena, offsets[n+1])."""
    offsets = np.zeros(len(signals) + 1, dtype=np.int64)
    if signals:
        offsets[1:] = np.cumsum([len(s) for s in signals])
    arena = np.empty(int(offsets[-1]), dtype=np.int16)
    for i, s in enumerate(signals):
        arena[offsets[i]:offsets[i + 1]] = s
    return arena, offsets


class BatchExCall:
    """One pxg_process_batch_ex call prepared for somebody else to make (NativeContext.batch_ex_call): the extras
    struct, the output arrays it points at, the function's address and the context handle.  After the call:
    result(rc) -- the dict process_batch_ex returns --, or None when a variable-size output outgrew its buffer
    (PXG_E_NOMEM with the totals set: the caller goes through process_batch_ex, whose loop sizes them)."""

    def __init__(self, ctx, n, stage_mask, unsplit, want_spikes):
        self.ctx, self.n, self.stage_mask = ctx, n, stage_mask
        self.function = ctx.__dict__.get('_batch_ex_address')
        if self.function is None:
            self.function = ctx._batch_ex_address = C.cast(ctx.lib.pxg_process_batch_ex, C.c_void_p).value
        self.handle = ctx.handle.value if hasattr(ctx.handle, 'value') else int(ctx.handle)
        x = self.x = PxgBatchExtras()
        x.struct_bytes = C.sizeof(PxgBatchExtras)
        self.records = np.zeros(n, dtype=RESULT_DTYPE)
        self.keep = []
        self.cnt = self.iv = self.rows = self.spike_off = None
        if unsplit is not None:
            first = np.ascontiguousarray(unsplit[0], dtype=np.int64)
            nb = np.ascontiguousarray(unsplit[1], dtype=np.int64)
            if len(first) != n or len(nb) != n:
                raise ValueError('one first_sample / n_blocks entry per read')
            self.keep += [first, nb]
            self.cnt = np.zeros(n, dtype=np.int32)
            x.unsplit_first_sample, x.unsplit_n_blocks = first.ctypes.data, nb.ctypes.data
            x.unsplit_block_stride = int(unsplit[2])
            x.unsplit_count = self.cnt.ctypes.data
            self.iv_cap = max(getattr(ctx, '_unsplit_cap', 0), n // 4 + 1024)
            self.iv = np.empty((self.iv_cap, 2), dtype=np.int64)
            x.unsplit_cap, x.unsplit_intervals = self.iv_cap, self.iv.ctypes.data
        if want_spikes:
            self.spike_cap = max(getattr(ctx, '_spike_cap', 0), 2 * n + 1024)
            self.spike_off = np.zeros(n + 1, dtype=np.int64)
            self.rows = np.zeros((self.spike_cap, 4), dtype=np.float32)
            x.spike_cap, x.spikes, x.spike_offsets = self.spike_cap, self.rows.ctypes.data, self.spike_off.ctypes.data
        self.extras = C.addressof(x)

    def result(self, rc):
        x = self.x
        if rc == PXG_E_NOMEM and ((self.iv is not None and x.unsplit_total > self.iv_cap) or
                                  (self.rows is not None and x.spike_total > self.spike_cap)):
            return None
        self.ctx._check(rc, 'pxg_process_batch_ex')
        res = {'records': self.records}
        if self.rows is not None:
            self.ctx._spike_cap = self.spike_cap
            res['spikes'] = (self.rows[:int(x.spike_total)], self.spike_off)
        if self.iv is not None:
            self.ctx._unsplit_cap = self.iv_cap
            start = np.zeros(self.n + 1, dtype=np.int64)
            np.cumsum(np.maximum(self.cnt, 0), out=start[1:])
            res['unsplit'] = (self.iv[:int(x.unsplit_total)], self.cnt, start)
        return res


class NativeContext:
    """One pxg_ctx: models resident on one GPU for the life of a worker
    (the role of WorkerPersistenceStorage, worker_persistence.py:46-90)."""

    def __init__(self, config, device_id=0):
        self.lib = load_library()
        self.ncfg = NativeConfig(config, device_id)
        self.cfg = self.ncfg.struct
        self.state_names = self.ncfg.state_names
        handle = C.c_void_p()
        rc = self.lib.pxg_create(C.byref(self.cfg), C.byref(handle))
        if rc != 0:
            raise PxgError('pxg_create failed ({}): {}'.format(
                rc, (self.lib.pxg_last_error(None) or b'').decode()))
        self.handle = handle
        self.n_resident = 0

    def close(self):
        if getattr(self, 'handle', None):
            for addr in list(getattr(self, '_pinned', {})):       # whatever is still page-locked through this context
                self.lib.pxg_host_unregister(self.handle, C.c_void_p(addr))
            self._pinned = {}
            self.lib.pxg_destroy(self.handle)
            self.handle = None

    __del__ = close

    def _check(self, rc, what):
        if rc != 0:
            raise PxgError('{} failed ({}): {}'.format(
                what, rc, (self.lib.pxg_last_error(self.handle) or b'').decode()))

    def device_info(self):
        info = PxgDeviceInfo()
        self._check(self.lib.pxg_get_device_info(self.handle, C.byref(info)), 'device_info')
        return {'name': info.name.decode(), 'arch': info.arch.decode(),
                'compute_units': info.compute_units, 'wavefront_size': info.wavefront_size,
                'total_mem': info.total_mem, 'lds_per_cu': info.lds_per_cu,
                'clock_khz': info.clock_khz}

    # ---- whole path ------------------------------------------------------
    @staticmethod
    def _prep(arena, offsets, calib, scale_shift):
        arena = np.ascontiguousarray(arena, dtype=np.int16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        calib = np.ascontiguousarray(calib, dtype=CALIB_DTYPE)
        n = len(offsets) - 1
        if len(calib) != n:
            raise ValueError('calib must have one row per read')
        if scale_shift is not None:
            scale_shift = np.ascontiguousarray(scale_shift, dtype=np.float32).reshape(n, 2)
        return arena, offsets, calib, scale_shift, n

    def process_batch(self, arena, offsets, calib, scale_shift=None,
                      stage_mask=STAGE_ALL_DEMUX):
        arena, offsets, calib, scale_shift, n = self._prep(arena, offsets, calib, scale_shift)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        self._check(self.lib.pxg_process_batch(
            self.handle, n, _ptr(arena), _ptr(offsets), _ptr(calib), _ptr(scale_shift),
            stage_mask, _ptr(out)), 'pxg_process_batch')
        return out

    def upload(self, arena, offsets, calib, scale_shift=None):
        arena, offsets, calib, scale_shift, n = self._prep(arena, offsets, calib, scale_shift)
        self._check(self.lib.pxg_batch_upload(
            self.handle, n, _ptr(arena), _ptr(offsets), _ptr(calib), _ptr(scale_shift)),
            'pxg_batch_upload')
        self.n_resident = n

    def upload_tiled(self, n_reads, arena, offsets, calib, scale_shift=None, phase=0):
        """Resident batch of `n_reads` reads, read j = base read (phase + j) % len(base),
        replicated on the device from ONE upload of the distinct base reads (configs[3]/[4]
        shapes: 12-15 GB of int16 per GPU never exist on the host)."""
        arena, offsets, calib, scale_shift, nb = self._prep(arena, offsets, calib, scale_shift)
        self._check(self.lib.pxg_batch_upload_tiled(
            self.handle, int(n_reads), nb, int(phase), _ptr(arena), _ptr(offsets), _ptr(calib),
            _ptr(scale_shift)), 'pxg_batch_upload_tiled')
        self.n_resident = int(n_reads)

    def stage(self, arena, offsets, calib, scale_shift=None, prefix_limit=0):
        """Copy the NEXT batch into the spare input slot on the copy stream while the
        resident batch computes; the arrays are kept alive until swap().  prefix_limit > 0: only that many
        samples of each read cross the link (include/pxg.h, pxg_batch_stage_prefix: enough for runs without
        poly(A) / chimera scan when it is prefix_limit_for(mask))."""
        arena, offsets, calib, scale_shift, n = self._prep(arena, offsets, calib, scale_shift)
        self._staged = (arena, offsets, calib, scale_shift, n)
        self._check(self.lib.pxg_batch_stage_prefix(
            self.handle, n, _ptr(arena), _ptr(offsets), _ptr(calib), _ptr(scale_shift), int(prefix_limit)),
            'pxg_batch_stage')

    def prefix_limit_for(self, stage_mask, whole_read_hooks=False):
        """Samples of a read the stages of `stage_mask` can reach (0 = all of them): without poly(A) and without
        the hooks that walk whole reads (chimera scan, event table) nothing reads behind the segmentation's scan
        limit (signal_analyzer.py:347-349)."""
        if (stage_mask & STAGE_POLYA) or whole_read_hooks:
            return 0
        return int(max(self.cfg.scaler_length, self.cfg.segmentation_scan_limit))

    def download_samples(self, n_samples):
        """The resident batch's int16 samples (what stage_z decoded on the device)."""
        out = np.empty(int(n_samples), dtype=np.int16)
        self._check(self.lib.pxg_batch_download_samples(self.handle, _ptr(out)), 'pxg_batch_download_samples')
        return out

    def stage_z(self, enc, offsets, calib, scale_shift=None, prefix_limit=0):
        """stage() for samples that arrive encoded (EncodedSamples): the bytes cross the link
        and are decoded on the device into the spare input slot (prefix_limit: as in stage())."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        calib = np.ascontiguousarray(calib, dtype=CALIB_DTYPE)
        n = len(offsets) - 1
        if scale_shift is not None:
            scale_shift = np.ascontiguousarray(scale_shift, dtype=np.float32).reshape(n, 2)
        z = np.ascontiguousarray(enc.z, dtype=np.uint8)
        chunks = np.ascontiguousarray(enc.chunks, dtype=Z_CHUNK_DTYPE)
        if int(offsets[-1]) != enc.n_samples:
            raise ValueError('offsets describe {} samples, the encoded slice {}'.format(int(offsets[-1]), enc.n_samples))
        self._staged = ((z, chunks), offsets, calib, scale_shift, n)
        self._check(self.lib.pxg_batch_stage_z_prefix(
            self.handle, n, _ptr(z), len(z), _ptr(chunks), len(chunks), enc.data_base, enc.dst_base,
            _ptr(offsets), _ptr(calib), _ptr(scale_shift), int(prefix_limit)), 'pxg_batch_stage_z')

    def swap(self):
        """Make the staged batch the resident one (waits for its copies)."""
        self._check(self.lib.pxg_batch_swap(self.handle), 'pxg_batch_swap')
        self.n_resident = self._staged[4]
        self._staged = None

    def pin(self, array):
        """Page-lock a NumPy array so stage() copies are DMA transfers; returns it."""
        self._check(self.lib.pxg_host_register(self.handle, _ptr(array), array.nbytes),
                    'pxg_host_register')
        self._pinned = getattr(self, '_pinned', {})
        self._pinned[_ptr(array).value] = array       # (kept alive until it is released)
        return array

    def unpin(self, array):
        """Release a page lock.  Works after close() too (page locks are process-global): a staging arena
        that outlives its context is released, not left registered on memory about to be unmapped."""
        getattr(self, '_pinned', {}).pop(_ptr(array).value, None)
        rc = self.lib.pxg_host_unregister(self.handle if getattr(self, 'handle', None) else None, _ptr(array))
        if rc != 0:
            raise PxgError('pxg_host_unregister failed ({})'.format(rc))

    def run(self, stage_mask=STAGE_ALL_DEMUX):
        self._check(self.lib.pxg_batch_run(self.handle, stage_mask), 'pxg_batch_run')

    def sync(self):
        self._check(self.lib.pxg_batch_sync(self.handle), 'pxg_batch_sync')

    def download(self, out=None):
        """Result records of the last run.  `out`: a caller-owned RESULT_DTYPE array of at
        least n_resident rows to receive them (page-lock it with pin() and reuse it across
        batches: the D2H copy is then a direct DMA transfer, not a staged one)."""
        if out is None:
            out = np.zeros(self.n_resident, dtype=RESULT_DTYPE)
        elif out.dtype != RESULT_DTYPE or len(out) < self.n_resident or not out.flags.c_contiguous:
            raise ValueError('out must be a contiguous RESULT_DTYPE array of >= n_resident rows')
        self._check(self.lib.pxg_batch_download(self.handle, _ptr(out)), 'pxg_batch_download')
        return out[:self.n_resident]

    def download_spikes(self, records=None):
        """Spike rows of the last run as (rows [total, 4] float32, offsets [n + 1] int64):
        rows[offsets[r]:offsets[r + 1]] are read r's spikes -- all of them (polya.py:109-115)."""
        n = self.n_resident
        off = np.zeros(n + 1, dtype=np.int64)
        total = 0
        if records is not None and len(records) == n:      # the records say how many rows there are
            total = int(np.where(records['polya_called'] != 0, records['polya_n_spikes'], 0).sum())
        while True:
            rows = np.zeros((total, 4), dtype=np.float32)
            rc = self.lib.pxg_batch_download_spikes(self.handle, total, _ptr(rows), _ptr(off))
            if rc == PXG_E_NOMEM and int(off[-1]) > total:
                total = int(off[-1])
                continue
            self._check(rc, 'pxg_batch_download_spikes')
            return rows, off

    def batch_ex_call(self, n, stage_mask=STAGE_ALL_DEMUX, unsplit=None, want_spikes=False):
        """The arguments of ONE pxg_process_batch_ex call over n reads with plain int16 samples, laid out for a caller
        that makes the call itself -- csrc/pxg_pyreport.c decode_and_run, which runs the FAST5 decode and this call behind
        one release of the interpreter lock.  See BatchExCall; process_batch_ex is the same call made from here."""
        return BatchExCall(self, n, stage_mask, unsplit, want_spikes)

    def process_batch_ex(self, samples, offsets, calib, stage_mask=STAGE_ALL_DEMUX, scale_shift=None,
                         unsplit=None, want_spikes=False):
        """pxg_process_batch_ex: ONE call per worker batch, callable from several threads at once
        (calls overlap on the device; the GIL is released for the whole call).  `samples`: an
        int16 arena or EncodedSamples.  `unsplit`: (first_sample, n_blocks, block_stride) to run
        the a19 window scan on the same resident batch.  Returns a dict: records, and when asked
        spikes = (rows, offsets), unsplit = (intervals, count, start)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        calib = np.ascontiguousarray(calib, dtype=CALIB_DTYPE)
        n = len(offsets) - 1
        if len(calib) != n:
            raise ValueError('calib must have one row per read')
        x = PxgBatchExtras()
        x.struct_bytes = C.sizeof(PxgBatchExtras)
        keep = [offsets, calib]
        if scale_shift is not None:
            scale_shift = np.ascontiguousarray(scale_shift, dtype=np.float32).reshape(n, 2)
            x.scale_shift_or_null = scale_shift.ctypes.data
            keep.append(scale_shift)
        arena = None
        if isinstance(samples, EncodedSamples):
            if int(offsets[-1]) != samples.n_samples:
                raise ValueError('offsets describe {} samples, the encoded slice {}'.format(
                    int(offsets[-1]), samples.n_samples))
            z = np.ascontiguousarray(samples.z, dtype=np.uint8)
            chunks = np.ascontiguousarray(samples.chunks, dtype=Z_CHUNK_DTYPE)
            keep += [z, chunks]
            x.z, x.z_bytes = z.ctypes.data if len(z) else None, len(z)
            x.chunks, x.n_chunks = chunks.ctypes.data if len(chunks) else None, len(chunks)
            x.data_base, x.dst_base = samples.data_base, samples.dst_base
            if not len(z):                                 # nothing encoded: an empty int16 arena
                arena = np.zeros(0, dtype=np.int16)
        else:
            arena = np.ascontiguousarray(samples, dtype=np.int16)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        cnt = iv = None
        if unsplit is not None:
            first = np.ascontiguousarray(unsplit[0], dtype=np.int64)
            nb = np.ascontiguousarray(unsplit[1], dtype=np.int64)
            if len(first) != n or len(nb) != n:
                raise ValueError('one first_sample / n_blocks entry per read')
            keep += [first, nb]
            cnt = np.zeros(n, dtype=np.int32)
            x.unsplit_first_sample, x.unsplit_n_blocks = first.ctypes.data, nb.ctypes.data
            x.unsplit_block_stride = int(unsplit[2])
            x.unsplit_count = cnt.ctypes.data
        iv_cap = max(getattr(self, '_unsplit_cap', 0), n // 4 + 1024) if unsplit is not None else 0
        spike_cap = max(getattr(self, '_spike_cap', 0), 2 * n + 1024) if want_spikes else 0
        spike_off = np.zeros(n + 1, dtype=np.int64) if want_spikes else None
        while True:
            if unsplit is not None:
                iv = np.empty((iv_cap, 2), dtype=np.int64)
                x.unsplit_cap, x.unsplit_intervals = iv_cap, iv.ctypes.data
            rows = None
            if want_spikes:
                rows = np.zeros((spike_cap, 4), dtype=np.float32)
                x.spike_cap, x.spikes, x.spike_offsets = spike_cap, rows.ctypes.data, spike_off.ctypes.data
            rc = self.lib.pxg_process_batch_ex(self.handle, n, _ptr(arena), _ptr(offsets), _ptr(calib),
                                               stage_mask, C.byref(x), _ptr(out))
            if rc == PXG_E_NOMEM and (x.unsplit_total > iv_cap or x.spike_total > spike_cap):
                # rare: a variable-size output outgrew the guess; the batch goes through again
                iv_cap = max(iv_cap, int(x.unsplit_total) + 1024) if unsplit is not None else 0
                spike_cap = max(spike_cap, int(x.spike_total) + 1024) if want_spikes else 0
                continue
            self._check(rc, 'pxg_process_batch_ex')
            break
        self._unsplit_cap, self._spike_cap = iv_cap, spike_cap
        res = {'records': out}
        if want_spikes:
            res['spikes'] = (rows[:int(x.spike_total)], spike_off)
        if unsplit is not None:
            start = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(np.maximum(cnt, 0), out=start[1:])
            res['unsplit'] = (iv[:int(x.unsplit_total)], cnt, start)
        return res

    def merge_stats(self):
        """(groups, calls): merged batches run so far and the small process_batch calls they carried
        (include/pxg.h, "small calls share a batch")."""
        g, c = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.pxg_merge_stats(self.handle, C.byref(g), C.byref(c)), 'pxg_merge_stats')
        return int(g.value), int(c.value)

    def stage_times(self):
        t = PxgStageTimes()
        self._check(self.lib.pxg_batch_times(self.handle, C.byref(t)), 'pxg_batch_times')
        return ({name: float(t.ms[i]) for i, name in enumerate(TIMER_NAMES)},
                {name: int(t.n_launches[i]) for i, name in enumerate(TIMER_NAMES)})

    # ---- stage hooks -----------------------------------------------------
    def raw_to_pa(self, raw, calib_row):
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        cal = np.ascontiguousarray(calib_row, dtype=CALIB_DTYPE).reshape(1)
        out = np.empty(len(raw), dtype=np.float32)
        self._check(self.lib.pxg_raw_to_pa(self.handle, len(raw), _ptr(raw), _ptr(cal),
                                           _ptr(out)), 'pxg_raw_to_pa')
        return out

    def head_pool(self, arena, offsets, calib):
        arena, offsets, calib, _, n = self._prep(arena, offsets, calib, None)
        width = self.cfg.scaler_length // self.cfg.stride
        out = np.zeros((n, width), dtype=np.float32)
        status = np.zeros(n, dtype=np.int32)
        self._check(self.lib.pxg_head_pool(self.handle, n, _ptr(arena), _ptr(offsets),
                                           _ptr(calib), _ptr(out), _ptr(status)),
                    'pxg_head_pool')
        return out, status

    def scaler_lstm(self, head):
        head = np.ascontiguousarray(head, dtype=np.float32)
        n = head.shape[0]
        if head.shape[1] != self.cfg.scaler_length // self.cfg.stride:
            raise ValueError('head must be n x (scaler_length/stride)')
        pred = np.zeros((n, 2), dtype=np.float32)
        self._check(self.lib.pxg_scaler_lstm(self.handle, n, _ptr(head), _ptr(pred)),
                    'pxg_scaler_lstm')
        return pred

    def scaler_transform(self, pred):
        pred = np.ascontiguousarray(pred, dtype=np.float32)
        n = pred.shape[0]
        ss = np.zeros((n, 2), dtype=np.float32)
        status = np.zeros(n, dtype=np.int32)
        self._check(self.lib.pxg_scaler_transform(self.handle, n, _ptr(pred), _ptr(ss),
                                                  _ptr(status)), 'pxg_scaler_transform')
        return ss, status

    def pool_scale(self, arena, offsets, calib, scale_shift):
        arena, offsets, calib, scale_shift, n = self._prep(arena, offsets, calib, scale_shift)
        lens = np.diff(offsets) // self.cfg.stride
        poff = np.zeros(n + 1, dtype=np.int64)
        poff[1:] = np.cumsum(lens)
        out = np.zeros(int(poff[-1]), dtype=np.float32)
        self._check(self.lib.pxg_pool_scale(self.handle, n, _ptr(arena), _ptr(offsets),
                                            _ptr(calib), _ptr(scale_shift), _ptr(poff),
                                            _ptr(out)), 'pxg_pool_scale')
        return out, poff

    def viterbi(self, signals, which_model=0, want_path=False):
        sigs = [np.ascontiguousarray(s, dtype=np.float32) for s in signals]
        n = len(sigs)
        off = np.zeros(n + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in sigs])
        arena = np.concatenate(sigs) if n else np.zeros(0, np.float32)
        first = np.zeros((n, PXG_N_SEGMENTS), dtype=np.int32)
        last = np.zeros((n, PXG_N_SEGMENTS), dtype=np.int32)
        path = np.zeros(int(off[-1]), dtype=np.int32) if want_path else None
        logp = np.zeros(n, dtype=np.float64)
        self._check(self.lib.pxg_viterbi(self.handle, which_model, n, _ptr(arena), _ptr(off),
                                         _ptr(first), _ptr(last), _ptr(path), _ptr(logp)),
                    'pxg_viterbi')
        paths = [path[off[i]:off[i + 1]] for i in range(n)] if want_path else None
        return first, last, paths, logp

    def barcode_window(self, signals):
        sigs = [np.ascontiguousarray(s, dtype=np.float32) for s in signals]
        n = len(sigs)
        off = np.zeros(n + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in sigs])
        arena = np.concatenate(sigs) if n else np.zeros(0, np.float32)
        out = np.zeros((n, self.cfg.signal_trim_length), dtype=np.float32)
        pushed = np.zeros(n, dtype=np.int8)
        self._check(self.lib.pxg_barcode_window(self.handle, n, _ptr(arena), _ptr(off),
                                                _ptr(out), _ptr(pushed)),
                    'pxg_barcode_window')
        return out, pushed

    def demux_lstm(self, win):
        win = np.ascontiguousarray(win, dtype=np.float32)
        n = win.shape[0]
        if win.shape[1] != self.cfg.signal_trim_length:
            raise ValueError('win must be n x signal_trim_length')
        probs = np.zeros((n, self.cfg.demux_dense.out_dim), dtype=np.float32)
        self._check(self.lib.pxg_demux_lstm(self.handle, n, _ptr(win), _ptr(probs)),
                    'pxg_demux_lstm')
        return probs

    def guppy_event_means(self, arena, offsets, calib, scale_shift, first_sample, n_blocks,
                          block_stride=15):
        arena, offsets, calib, scale_shift, n = self._prep(arena, offsets, calib, scale_shift)
        first = np.ascontiguousarray(first_sample, dtype=np.int64)
        eoff = np.zeros(n + 1, dtype=np.int64)
        eoff[1:] = np.cumsum(n_blocks)
        mean = np.zeros(int(eoff[-1]), dtype=np.float32)
        scaled = np.zeros(int(eoff[-1]), dtype=np.float32)
        self._check(self.lib.pxg_guppy_event_means(
            self.handle, n, _ptr(arena), _ptr(offsets), _ptr(calib), _ptr(scale_shift),
            _ptr(first), _ptr(eoff), block_stride, _ptr(mean), _ptr(scaled)),
            'pxg_guppy_event_means')
        return mean, scaled, eoff

    def unsplit_scan(self, first_sample, n_blocks, block_stride=15):
        """Window scan of detect_unsplit_read on the resident batch.  Returns (intervals
        [total, 2] int64, count [n] int32, start [n+1] int64): the candidates of read r are
        intervals[start[r]:start[r+1]]; count[r] < 0 is that read's own error code."""
        n = self.n_resident
        first = np.ascontiguousarray(first_sample, dtype=np.int64)
        nb = np.ascontiguousarray(n_blocks, dtype=np.int64)
        if len(first) != n or len(nb) != n:
            raise ValueError('one first_sample / n_blocks entry per resident read')
        cnt = np.empty(n, dtype=np.int32)
        total = C.c_int64(0)
        cap = max(getattr(self, '_unsplit_cap', 0), n // 4 + 1024)
        while True:
            iv = np.empty((cap, 2), dtype=np.int64)
            self._check(self.lib.pxg_batch_unsplit_scan(self.handle, _ptr(first), _ptr(nb),
                                                        block_stride, cap, _ptr(iv), _ptr(cnt),
                                                        C.byref(total)), 'pxg_batch_unsplit_scan')
            if total.value <= cap:
                break
            cap = int(total.value) + 1024        # rare: more candidates than the guess, run again
        self._unsplit_cap = cap
        start = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.maximum(cnt, 0), out=start[1:])
        return iv[:total.value], cnt, start

    def unsplit_scan_events(self, n_events, ev_start, ev_mean):
        """The same scan for reads whose basecall brings its own event table (albacore `Events'):
        n_events [n] rows per resident read (0 leaves a read out), their ascending `start' column (int64
        samples) and float32 `mean' column back to back.  Returns what unsplit_scan returns."""
        n = self.n_resident
        ne = np.ascontiguousarray(n_events, dtype=np.int64)
        st = np.ascontiguousarray(ev_start, dtype=np.int64)
        mean = np.ascontiguousarray(ev_mean, dtype=np.float32)
        total_ev = int(np.maximum(ne, 0).sum())
        if len(ne) != n or len(st) != total_ev or len(mean) != total_ev:
            raise ValueError('one n_events entry per resident read, one start / mean per event')
        cnt = np.empty(n, dtype=np.int32)
        total = C.c_int64(0)
        cap = max(getattr(self, '_unsplit_cap', 0), n // 4 + 1024)
        while True:
            iv = np.empty((cap, 2), dtype=np.int64)
            self._check(self.lib.pxg_batch_unsplit_scan_events(self.handle, _ptr(ne), _ptr(st), _ptr(mean), cap,
                                                               _ptr(iv), _ptr(cnt), C.byref(total)),
                        'pxg_batch_unsplit_scan_events')
            if total.value <= cap:
                break
            cap = int(total.value) + 1024
        self._unsplit_cap = cap
        start = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.maximum(cnt, 0), out=start[1:])
        return iv[:total.value], cnt, start

    def lock(self, which):
        """pxg_ctx_lock: 0 = the spare input slot, 1 = the resident batch (the locks pxg_process_batch_ex
        takes inside); blocks with the GIL released."""
        self._check(self.lib.pxg_ctx_lock(self.handle, int(which)), 'pxg_ctx_lock')

    def unlock(self, which):
        self._check(self.lib.pxg_ctx_unlock(self.handle, int(which)), 'pxg_ctx_unlock')

    def download_windows(self, records=None):
        """[n, signal_trim_length] float32: the classifier's input windows of the resident batch
        (rows of reads that were not pushed are zeroed when `records` is given)."""
        out = np.empty((self.n_resident, self.cfg.signal_trim_length), dtype=np.float32)
        self._check(self.lib.pxg_batch_download_windows(self.handle, _ptr(out)), 'pxg_batch_download_windows')
        if records is not None:
            out[records['bc_pushed'] == 0] = 0.0
        return out

    def event_table(self, first_sample, n_blocks, block_stride=15):
        """(mean, stdv, scaled_mean, offsets [n + 1]) of the Guppy blocks of the resident reads
        (n_blocks 0 leaves a read out): the numeric columns of the dumped event table."""
        n = self.n_resident
        first = np.ascontiguousarray(first_sample, dtype=np.int64)
        if len(first) != n or len(n_blocks) != n:
            raise ValueError('one first_sample / n_blocks entry per resident read')
        offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.maximum(np.asarray(n_blocks, dtype=np.int64), 0), out=offsets[1:])
        mean, stdv, scaled = (np.empty(int(offsets[-1]), dtype=np.float32) for _ in range(3))
        self._check(self.lib.pxg_batch_event_table(self.handle, _ptr(first), _ptr(offsets), int(block_stride),
                                                   _ptr(mean), _ptr(stdv), _ptr(scaled)), 'pxg_batch_event_table')
        return mean, stdv, scaled, offsets

    def pooled_signal(self, first, count):
        """load_signal(pool=stride)[first[r] : first[r] + count[r]] of every resident read with the
        scale / shift of the last run (count 0 leaves a read out).  Returns (values float32,
        offsets [n + 1])."""
        n = self.n_resident
        first = np.ascontiguousarray(first, dtype=np.int64)
        offsets = np.zeros(n + 1, dtype=np.int64)
        if len(first) != n or len(count) != n:
            raise ValueError('one first / count entry per resident read')
        np.cumsum(np.maximum(np.asarray(count, dtype=np.int64), 0), out=offsets[1:])
        out = np.empty(int(offsets[-1]), dtype=np.float32)
        self._check(self.lib.pxg_batch_pooled_signal(self.handle, _ptr(first), _ptr(offsets), _ptr(out)),
                    'pxg_batch_pooled_signal')
        return out, offsets

    def polya(self, arena, offsets, calib, scale_shift, seg_first, seg_last, want_spikes=True):
        """a14-a17 on caller-supplied scaling and segmentation (standalone hook)."""
        arena, offsets, calib, scale_shift, n = self._prep(arena, offsets, calib, scale_shift)
        sf = np.ascontiguousarray(seg_first, dtype=np.int32).reshape(n, PXG_N_SEGMENTS)
        sl = np.ascontiguousarray(seg_last, dtype=np.int32).reshape(n, PXG_N_SEGMENTS)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        off = np.zeros(n + 1, dtype=np.int64) if want_spikes else None
        cap = 2 * n + 64
        while True:
            rows = np.zeros((cap, 4), dtype=np.float32) if want_spikes else None
            rc = self.lib.pxg_polya(self.handle, n, _ptr(arena), _ptr(offsets), _ptr(calib),
                                    _ptr(scale_shift), _ptr(sf), _ptr(sl), _ptr(out), cap, _ptr(rows), _ptr(off))
            if rc == PXG_E_NOMEM and want_spikes and int(off[-1]) > cap:
                cap = int(off[-1])
                continue
            self._check(rc, 'pxg_polya')
            break
        return out, ((rows[:int(off[-1])], off) if want_spikes else None)

    def detect_events(self, signals, max_events=None):
        sigs = [np.ascontiguousarray(s, dtype=np.float32) for s in signals]
        if any(s.ndim != 1 for s in sigs):
            raise ValueError('Expects an 1-dimensional array.')   # csupport.c:97-101
        n = len(sigs)
        off = np.zeros(n + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in sigs])
        arena = np.concatenate(sigs) if n else np.zeros(0, np.float32)
        cap = int(max_events or (max(len(s) for s in sigs) // 4 + 2))
        ev = np.zeros((n, cap), dtype=EVENT_DTYPE)
        cnt = np.zeros(n, dtype=np.int64)
        self._check(self.lib.pxg_detect_events(self.handle, n, _ptr(arena), _ptr(off), cap,
                                               _ptr(ev), _ptr(cnt)), 'pxg_detect_events')
        return [ev[i, :min(cnt[i], cap)] for i in range(n)], cnt
