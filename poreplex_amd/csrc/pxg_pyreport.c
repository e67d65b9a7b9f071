/*
 * pxg_pyreport.c -- result dicts of a worker batch, built from the batch table's columns.
 *
 * The reference hands `process_batch`'s caller one dict per read (NanoporeRead.report,
 * poreplex/signal_loader.py:165-198: keys in that order, optional keys only when set).  A GPU
 * worker batch is ten thousand reads, and building those dicts in Python (ReadTable.report)
 * was the largest host cost of the reference-shaped call (3.4 us per read; the whole GPU pass is
 * 1.5 us per read).  This CPython extension builds the SAME objects -- same keys, same key
 * order, same value types (str / int / float, `round(start / rate, 3)` correctly rounded the way
 * float.__round__ is, the int 0 for a missing mean_qscore, the (sequence, qstring, trim) tuple,
 * the poly(A) dict of polya.py:116-121) -- straight from the NumPy columns.  Host only; the
 * product works without it (ReadTable.report falls back to the Python loop, e.g. under another
 * interpreter version), tests/test_facade.py compares the two on randomised tables.
 *
 *   _pxgpy.report(columns: dict, rows: int64 buffer) -> list of dict
 *   _pxgpy.report_run(bundle columns: dict, first, n, records, adapter, barcoding, min_seq_len,
 *                     status_names, label_names[, measure_polya, spike rows, spike offsets, skip]) -> list of dict
 *
 * report_run is the whole host side of the USUAL worker call behind the GPU pass -- consecutive reads of a read
 * bundle, every one of them with a regular basecall summary (signal_analyzer.SignalAnalyzer.process_plain_run checks
 * that before the pass) --: the status / label rules of SignalAnalysis.process (signal_analyzer.py:230-286) and of
 * BarcodeDemultiplexer.predict (barcoding.py:108-118) applied to the pxg_read_result records, and the result dicts
 * built from the bundle's own columns, in one pass per read and without a batch table in between.  The general path
 * (ReadTable + SignalAnalyzer.judge + report) stays the definition; tests/test_plain_run.py holds the two against
 * each other on the golden batches and on randomised ones.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

#include "pxg.h"                  /* pxg_read_result, enum pxg_status: the records report_run reads */

#if PY_VERSION_HEX >= 0x030d0000 || !defined(_PyDict_NewPresized)
PyAPI_FUNC(PyObject*) _PyDict_NewPresized(Py_ssize_t minused);   /* exported by libpython, not in every public header */
#endif

typedef struct {
    Py_buffer view;
    int held;
} Buf;

static int get_buf(PyObject* cols, const char* name, Buf* b, Py_ssize_t itemsize, int optional)
{
    b->held = 0;
    memset(&b->view, 0, sizeof(b->view));
    PyObject* o = PyDict_GetItemString(cols, name);          /* borrowed */
    if (!o || o == Py_None) {
        if (optional) return 0;
        PyErr_Format(PyExc_KeyError, "report: column '%s' is missing", name);
        return -1;
    }
    if (PyObject_GetBuffer(o, &b->view, PyBUF_C_CONTIGUOUS) < 0) return -1;
    b->held = 1;
    if (b->view.itemsize != itemsize) {
        PyErr_Format(PyExc_TypeError, "report: column '%s' has item size %zd, expected %zd", name,
                     b->view.itemsize, itemsize);
        return -1;
    }
    return 0;
}

static PyObject* get_list(PyObject* cols, const char* name, int optional)
{
    PyObject* o = PyDict_GetItemString(cols, name);
    if (!o || o == Py_None) {
        if (!optional) PyErr_Format(PyExc_KeyError, "report: column '%s' is missing", name);
        return NULL;
    }
    if (!PyList_Check(o) && !PyTuple_Check(o)) {
        PyErr_Format(PyExc_TypeError, "report: column '%s' must be a list or tuple", name);
        return NULL;
    }
    return o;
}

/* round(x, 3) as float.__round__ does it: correctly rounded decimal, both ways */
static PyObject* round3(double x)
{
    if (!(x == x) || x - x != 0.0) return PyFloat_FromDouble(x);       /* nan / inf unchanged */
    char* s = PyOS_double_to_string(x, 'f', 3, 0, NULL);
    if (!s) return NULL;
    double r = PyOS_string_to_double(s, NULL, NULL);
    PyMem_Free(s);
    if (r == -1.0 && PyErr_Occurred()) return NULL;
    return PyFloat_FromDouble(r);
}

#define SET(d, key, value)                                  \
    do {                                                    \
        PyObject* v_ = (value);                             \
        if (!v_ || PyDict_SetItem(d, key, v_) < 0) {        \
            Py_XDECREF(v_);                                 \
            goto fail_row;                                  \
        }                                                   \
        Py_DECREF(v_);                                      \
    } while (0)
#define SET_BORROWED(d, key, value)                         \
    do {                                                    \
        if (PyDict_SetItem(d, key, (value)) < 0) goto fail_row; \
    } while (0)

enum { K_FILENAME, K_READ_ID, K_STATUS, K_CHANNEL, K_START_TIME, K_RUN_ID, K_SAMPLE_ID, K_DURATION,
       K_NUM_EVENTS, K_SEQUENCE_LENGTH, K_MEAN_QSCORE, K_SEQUENCE, K_ERROR_MESSAGE, K_LABEL, K_BARCODE,
       K_BARCODE_GUESS, K_BARCODE_SCORE, K_POLYA, K_BEGIN, K_END, K_DWELL_TIME, K_SPIKES, N_KEYS };
static const char* const KEY_NAMES[N_KEYS] = {
    "filename", "read_id", "status", "channel", "start_time", "run_id", "sample_id", "duration",
    "num_events", "sequence_length", "mean_qscore", "sequence", "error_message", "label", "barcode",
    "barcode_guess", "barcode_score", "polya", "begin", "end", "dwell_time", "spikes" };
static PyObject* KEYS[N_KEYS];

static PyObject* item(PyObject* seq, Py_ssize_t i)         /* borrowed; list or tuple */
{
    if (PyList_Check(seq)) {
        if (i < 0 || i >= PyList_GET_SIZE(seq)) { PyErr_SetString(PyExc_IndexError, "report: row outside a column"); return NULL; }
        return PyList_GET_ITEM(seq, i);
    }
    if (i < 0 || i >= PyTuple_GET_SIZE(seq)) { PyErr_SetString(PyExc_IndexError, "report: row outside a column"); return NULL; }
    return PyTuple_GET_ITEM(seq, i);
}

/* the dict of polya.py:116-121 from a called tail: begin, end, dwell time, and one (length, before, spike, after) tuple per
 * spike row; new reference, NULL with an exception set */
static PyObject* polya_dict(long long begin, long long end, double dwell_time, Py_ssize_t ns, const float* rows)
{
    PyObject* p = PyDict_New();
    if (!p) return NULL;
    PyObject* v;
    int bad = 0;
    bad |= !(v = PyLong_FromLongLong(begin)) || PyDict_SetItem(p, KEYS[K_BEGIN], v) < 0;
    Py_XDECREF(v);
    bad |= !(v = PyLong_FromLongLong(end)) || PyDict_SetItem(p, KEYS[K_END], v) < 0;
    Py_XDECREF(v);
    bad |= !(v = PyFloat_FromDouble(dwell_time)) || PyDict_SetItem(p, KEYS[K_DWELL_TIME], v) < 0;
    Py_XDECREF(v);
    PyObject* lst = PyList_New(ns);
    bad |= !lst;
    for (Py_ssize_t s = 0; lst && s < ns; s++) {
        const float* row = rows + (size_t)s * 4;
        PyObject* t = PyTuple_New(4);            /* (no format string to parse per spike: a fifth of the call) */
        if (!t) { bad = 1; break; }
        for (int c = 0; c < 4; c++) {
            PyObject* f = PyFloat_FromDouble((double)row[c]);
            if (!f) { bad = 1; break; }
            PyTuple_SET_ITEM(t, c, f);
        }
        if (bad) { Py_DECREF(t); break; }
        PyObject_GC_UnTrack(t);                  /* four floats */
        PyList_SET_ITEM(lst, s, t);
    }
    if (lst && !bad) bad |= PyDict_SetItem(p, KEYS[K_SPIKES], lst) < 0;
    Py_XDECREF(lst);
    if (bad) { Py_DECREF(p); return NULL; }
    /* dict -> (empty) list: nothing a cycle can run through as built, and a dict tracks itself again the moment
     * somebody stores a container in it.  Dicts that hold a NON-empty spike list stay tracked: the list is, and a
     * caller who ties it into a cycle must still be able to have that cycle collected. */
    if (ns == 0) PyObject_GC_UnTrack(p);
    return p;
}

static PyObject* report(PyObject* self, PyObject* args)
{
    PyObject *cols, *rows_obj;
    if (!PyArg_ParseTuple(args, "O!O", &PyDict_Type, &cols, &rows_obj)) return NULL;
    Buf rows, status, start_time, rate, duration, n_events, seq_len, qscore, has_summary, label, has_bc,
        barcode, guess, phred, seq_lazy, bundle_index, seq_arena, qual_arena, seq_off, polya_lazy, pa_begin,
        pa_end, pa_dwell, pa_nspk, spikes, spike_off, gpu_row;
    Buf* all[] = { &rows, &status, &start_time, &rate, &duration, &n_events, &seq_len, &qscore, &has_summary,
                   &label, &has_bc, &barcode, &guess, &phred, &seq_lazy, &bundle_index, &seq_arena,
                   &qual_arena, &seq_off, &polya_lazy, &pa_begin, &pa_end, &pa_dwell, &pa_nspk, &spikes,
                   &spike_off, &gpu_row };
    for (size_t k = 0; k < sizeof(all) / sizeof(all[0]); k++) all[k]->held = 0;
    PyObject* out = NULL;
    rows.held = 0;
    if (PyObject_GetBuffer(rows_obj, &rows.view, PyBUF_C_CONTIGUOUS) < 0) return NULL;
    rows.held = 1;
    if (rows.view.itemsize != 8) { PyErr_SetString(PyExc_TypeError, "report: rows must be int64"); goto done; }
    if (get_buf(cols, "status", &status, 1, 0) || get_buf(cols, "start_time", &start_time, 8, 0) ||
        get_buf(cols, "sampling_rate", &rate, 8, 0) || get_buf(cols, "duration", &duration, 8, 0) ||
        get_buf(cols, "num_events", &n_events, 8, 0) || get_buf(cols, "sequence_length", &seq_len, 8, 0) ||
        get_buf(cols, "mean_qscore", &qscore, 8, 0) || get_buf(cols, "has_summary", &has_summary, 1, 0) ||
        get_buf(cols, "label", &label, 1, 0) || get_buf(cols, "has_barcode", &has_bc, 1, 0) ||
        get_buf(cols, "barcode", &barcode, 1, 0) || get_buf(cols, "barcode_guess", &guess, 1, 0) ||
        get_buf(cols, "barcode_phred", &phred, 2, 0) || get_buf(cols, "seq_lazy", &seq_lazy, 1, 0) ||
        get_buf(cols, "bundle_index", &bundle_index, 8, 0) || get_buf(cols, "seq_arena", &seq_arena, 1, 1) ||
        get_buf(cols, "qual_arena", &qual_arena, 1, 1) || get_buf(cols, "seq_offsets", &seq_off, 8, 1) ||
        get_buf(cols, "polya_lazy", &polya_lazy, 1, 0) || get_buf(cols, "polya_begin", &pa_begin, 8, 0) ||
        get_buf(cols, "polya_end", &pa_end, 8, 0) || get_buf(cols, "polya_dwell_time", &pa_dwell, 8, 0) ||
        get_buf(cols, "polya_spike_count", &pa_nspk, 4, 0) || get_buf(cols, "spikes", &spikes, 4, 1) ||
        get_buf(cols, "spike_offsets", &spike_off, 8, 1) ||
        get_buf(cols, "gpu_row", &gpu_row, 8, 0))
        goto done;
    PyObject *filename = get_list(cols, "filename", 0), *read_id = get_list(cols, "read_id", 0),
             *channel = get_list(cols, "channel", 0), *run_id = get_list(cols, "run_id", 0),
             *sample_id = get_list(cols, "sample_id", 0), *sequence = get_list(cols, "sequence", 0),
             *error_message = get_list(cols, "error_message", 0), *polya = get_list(cols, "polya", 0),
             *status_names = get_list(cols, "status_names", 0), *label_names = get_list(cols, "label_names", 0);
    if (!filename || !read_id || !channel || !run_id || !sample_id || !sequence || !error_message || !polya ||
        !status_names || !label_names)
        goto done;
    {
        const Py_ssize_t n_rows = rows.view.len / 8;
        /* rows of the table = the SHORTEST per-row column: every one of them is indexed with the row number */
        Py_ssize_t n_table = status.view.len;                   /* int8 column: one byte per row */
        {
            Buf* per_row[] = { &start_time, &rate, &duration, &n_events, &seq_len, &qscore, &has_summary, &label, &has_bc,
                               &barcode, &guess, &phred, &seq_lazy, &bundle_index, &polya_lazy, &pa_begin, &pa_end,
                               &pa_dwell, &pa_nspk, &gpu_row };
            for (size_t c = 0; c < sizeof(per_row) / sizeof(per_row[0]); c++) {
                const Py_ssize_t rows_c = per_row[c]->view.len / (per_row[c]->view.itemsize > 0 ? per_row[c]->view.itemsize : 1);
                if (rows_c < n_table) n_table = rows_c;
            }
        }
        const int64_t* R = (const int64_t*)rows.view.buf;
        const int8_t* st = (const int8_t*)status.view.buf;
        const Py_ssize_t n_seq_off = seq_off.held ? seq_off.view.len / 8 : 0;
        /* spike rows of all GPU records back to back (CSR): rows of record g = [off[g], off[g + 1]) */
        const Py_ssize_t spike_rows = spikes.held && spike_off.held ? spikes.view.len / 16 : 0;
        const Py_ssize_t n_spike_off = spike_off.held ? spike_off.view.len / 8 : 0;
        out = PyList_New(n_rows);
        if (!out) goto done;
        for (Py_ssize_t k = 0; k < n_rows; k++) {
            const int64_t i = R[k];
            PyObject* d = NULL;
            if (i < 0 || i >= n_table) { PyErr_SetString(PyExc_IndexError, "report: row outside the table"); goto fail_row; }
            d = _PyDict_NewPresized(20);          /* 11 - 18 keys: no rehash on the way */
            if (!d) goto fail_row;
            PyObject* o;
            if (!(o = item(filename, i))) goto fail_row;
            SET_BORROWED(d, KEYS[K_FILENAME], o);
            if (!(o = item(read_id, i))) goto fail_row;
            SET_BORROWED(d, KEYS[K_READ_ID], o);
            if (!(o = item(status_names, st[i]))) goto fail_row;
            SET_BORROWED(d, KEYS[K_STATUS], o);
            if (!(o = item(channel, i))) goto fail_row;
            SET_BORROWED(d, KEYS[K_CHANNEL], o);
            {
                const double b = ((const double*)rate.view.buf)[i];
                if (b == 0.0) { PyErr_SetString(PyExc_ZeroDivisionError, "float division by zero"); goto fail_row; }
                SET(d, KEYS[K_START_TIME], round3((double)((const int64_t*)start_time.view.buf)[i] / b));
            }
            if (!(o = item(run_id, i))) goto fail_row;
            SET_BORROWED(d, KEYS[K_RUN_ID], o);
            if (!(o = item(sample_id, i))) goto fail_row;
            SET_BORROWED(d, KEYS[K_SAMPLE_ID], o);
            SET(d, KEYS[K_DURATION], PyLong_FromLongLong(((const int64_t*)duration.view.buf)[i]));
            SET(d, KEYS[K_NUM_EVENTS], PyLong_FromLongLong(((const int64_t*)n_events.view.buf)[i]));
            SET(d, KEYS[K_SEQUENCE_LENGTH], PyLong_FromLongLong(((const int64_t*)seq_len.view.buf)[i]));
            if (((const uint8_t*)has_summary.view.buf)[i])
                SET(d, KEYS[K_MEAN_QSCORE], PyFloat_FromDouble(((const double*)qscore.view.buf)[i]));
            else
                SET(d, KEYS[K_MEAN_QSCORE], PyLong_FromLong(0));
            /* sequence: the stored tuple, or (lazily) the bundle's text */
            if (!(o = item(sequence, i))) goto fail_row;
            if (o != Py_None) {
                SET_BORROWED(d, KEYS[K_SEQUENCE], o);
            } else if (((const uint8_t*)seq_lazy.view.buf)[i]) {
                const int64_t b = ((const int64_t*)bundle_index.view.buf)[i];
                if (!seq_arena.held || !qual_arena.held || b < 0 || b + 1 >= n_seq_off) {
                    PyErr_SetString(PyExc_IndexError, "report: lazy sequence outside the bundle");
                    goto fail_row;
                }
                const int64_t lo = ((const int64_t*)seq_off.view.buf)[b], hi = ((const int64_t*)seq_off.view.buf)[b + 1];
                if (lo < 0 || hi < lo || hi > seq_arena.view.len || hi > qual_arena.view.len) {
                    PyErr_SetString(PyExc_IndexError, "report: sequence offsets outside the arena");
                    goto fail_row;
                }
                PyObject* s = PyUnicode_DecodeASCII((const char*)seq_arena.view.buf + lo, hi - lo, NULL);
                PyObject* q = s ? PyUnicode_DecodeASCII((const char*)qual_arena.view.buf + lo, hi - lo, NULL) : NULL;
                PyObject* z = q ? PyLong_FromLong(0) : NULL;
                PyObject* t = z ? PyTuple_Pack(3, s, q, z) : NULL;
                Py_XDECREF(s); Py_XDECREF(q); Py_XDECREF(z);
                /* (str, str, int) cannot be part of a cycle: what the collector itself would find out on
                 * its first pass.  An untracked tuple also leaves the read's dict untracked, so a batch
                 * of plain results costs the cyclic collector nothing at all. */
                if (t) PyObject_GC_UnTrack(t);
                SET(d, KEYS[K_SEQUENCE], t);
            }
            if (!(o = item(error_message, i))) goto fail_row;
            {
                const int truth = PyObject_IsTrue(o);
                if (truth < 0) goto fail_row;
                if (truth) SET_BORROWED(d, KEYS[K_ERROR_MESSAGE], o);
            }
            {
                const int8_t lb = ((const int8_t*)label.view.buf)[i];
                if (lb != -1) {
                    if (!(o = item(label_names, lb))) goto fail_row;
                    SET_BORROWED(d, KEYS[K_LABEL], o);
                }
            }
            if (((const uint8_t*)has_bc.view.buf)[i]) {
                SET(d, KEYS[K_BARCODE], PyLong_FromLong(((const int8_t*)barcode.view.buf)[i]));
                SET(d, KEYS[K_BARCODE_GUESS], PyLong_FromLong(((const int8_t*)guess.view.buf)[i]));
                SET(d, KEYS[K_BARCODE_SCORE], PyLong_FromLong(((const int16_t*)phred.view.buf)[i]));
            }
            if (!(o = item(polya, i))) goto fail_row;
            if (o != Py_None) {
                SET_BORROWED(d, KEYS[K_POLYA], o);
            } else if (((const uint8_t*)polya_lazy.view.buf)[i]) {
                Py_ssize_t ns = ((const int32_t*)pa_nspk.view.buf)[i];
                const int64_t g = ((const int64_t*)gpu_row.view.buf)[i];
                int64_t at = 0;
                if (!spike_rows || g < 0 || g + 1 >= n_spike_off) ns = 0;
                else at = ((const int64_t*)spike_off.view.buf)[g];
                if (ns < 0 || at < 0 || at + ns > spike_rows) {
                    PyErr_SetString(PyExc_IndexError, "report: spike rows outside the table");
                    goto fail_row;
                }
                SET(d, KEYS[K_POLYA], polya_dict(((const int64_t*)pa_begin.view.buf)[i], ((const int64_t*)pa_end.view.buf)[i],
                                                 ((const double*)pa_dwell.view.buf)[i], ns,
                                                 ns ? (const float*)spikes.view.buf + (size_t)at * 4 : NULL));
                if (ns == 0) PyObject_GC_UnTrack(d);      /* (see polya_dict: nothing in it a cycle can run through) */
            }
            PyList_SET_ITEM(out, k, d);
            continue;
        fail_row:
            Py_XDECREF(d);
            Py_CLEAR(out);
            goto done;
        }
    }
done:
    for (size_t k = 0; k < sizeof(all) / sizeof(all[0]); k++)
        if (all[k]->held) PyBuffer_Release(&all[k]->view);
    return out;
}


/* a str from one element of a NumPy '<U' column: the characters before the trailing NULs (what ndarray.tolist() gives) */
static PyObject* ucs4_item(const Buf* col, Py_ssize_t i)
{
    const Py_ssize_t width = col->view.itemsize / 4;
    const Py_UCS4* p = (const Py_UCS4*)((const char*)col->view.buf + i * col->view.itemsize);
    Py_ssize_t len = width;
    while (len > 0 && p[len - 1] == 0) len--;
    return PyUnicode_FromKindAndData(PyUnicode_4BYTE_KIND, p, len);
}

static int get_ucs4(PyObject* cols, const char* name, Buf* b, Py_ssize_t n_reads)
{
    b->held = 0;
    memset(&b->view, 0, sizeof(b->view));
    PyObject* o = PyDict_GetItemString(cols, name);
    if (!o) { PyErr_Format(PyExc_KeyError, "report_run: column '%s' is missing", name); return -1; }
    if (PyObject_GetBuffer(o, &b->view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) < 0) return -1;
    b->held = 1;
    /* NumPy exports '<U8' as format "8w" (UCS4 code points) */
    if (!b->view.format || !strchr(b->view.format, 'w') || b->view.itemsize % 4 || b->view.itemsize <= 0 ||
        b->view.len / b->view.itemsize < n_reads) {
        PyErr_Format(PyExc_TypeError, "report_run: column '%s' must be a contiguous unicode array with a row per read", name);
        return -1;
    }
    return 0;
}

static int need_rows(const Buf* b, const char* name, Py_ssize_t rows)
{
    if (b->view.len / b->view.itemsize >= rows) return 0;
    PyErr_Format(PyExc_IndexError, "report_run: column '%s' is shorter than the bundle", name);
    return -1;
}

static PyObject* report_run(PyObject* self, PyObject* args)
{
    PyObject *cols, *rec_obj, *status_names, *label_names, *spikes_obj = Py_None, *spike_off_obj = Py_None, *skip_obj = Py_None,
             *short_obj = Py_None;
    Py_ssize_t first, n;
    int adapter, barcoding, polya = 0;
    long long min_seq_len;
    if (!PyArg_ParseTuple(args, "O!nnOipLO!O!|pOOOO", &PyDict_Type, &cols, &first, &n, &rec_obj, &adapter, &barcoding,
                          &min_seq_len, &PyTuple_Type, &status_names, &PyTuple_Type, &label_names, &polya, &spikes_obj,
                          &spike_off_obj, &skip_obj, &short_obj))
        return NULL;
    if (first < 0 || n < 0 || adapter < 0 || adapter >= PXG_N_SEGMENTS || PyTuple_GET_SIZE(status_names) < PXG_N_STATUS ||
        PyTuple_GET_SIZE(label_names) < 2) {
        PyErr_SetString(PyExc_ValueError, "report_run: bad arguments");
        return NULL;
    }
    Buf rec, start_time, duration, calib, present, seq_len, qscore, n_events, seq_off, seq_arena, qual_arena, channel,
        run_id, sample_id, spikes, spike_off, skip, too_short;
    Buf* all[] = { &rec, &start_time, &duration, &calib, &present, &seq_len, &qscore, &n_events, &seq_off, &seq_arena,
                   &qual_arena, &channel, &run_id, &sample_id, &spikes, &spike_off, &skip, &too_short };
    for (size_t k = 0; k < sizeof(all) / sizeof(all[0]); k++) all[k]->held = 0;
    PyObject* out = NULL;
    const Py_ssize_t last = first + n;         /* reads [first, last) of the bundle */
    if (PyObject_GetBuffer(rec_obj, &rec.view, PyBUF_C_CONTIGUOUS) < 0) return NULL;
    rec.held = 1;
    if (rec.view.itemsize != (Py_ssize_t)sizeof(pxg_read_result) || rec.view.len / rec.view.itemsize < n) {
        PyErr_SetString(PyExc_TypeError, "report_run: records must be pxg_read_result items, one per read");
        goto done;
    }
    if (get_buf(cols, "start_time", &start_time, 8, 0) || get_buf(cols, "duration", &duration, 8, 0) ||
        get_buf(cols, "calib", &calib, 32, 0) || get_buf(cols, "bc_present", &present, 1, 0) ||
        get_buf(cols, "bc_sequence_length", &seq_len, 8, 0) || get_buf(cols, "bc_mean_qscore", &qscore, 8, 0) ||
        get_buf(cols, "bc_num_events", &n_events, 8, 0) || get_buf(cols, "seq_offsets", &seq_off, 8, 0) ||
        get_buf(cols, "seq_arena", &seq_arena, 1, 0) || get_buf(cols, "qual_arena", &qual_arena, 1, 0) ||
        get_ucs4(cols, "channel_number", &channel, last) || get_ucs4(cols, "run_id", &run_id, last) ||
        get_ucs4(cols, "sample_id", &sample_id, last))
        goto done;
    if (need_rows(&start_time, "start_time", last) || need_rows(&duration, "duration", last) ||
        need_rows(&calib, "calib", last) || need_rows(&present, "bc_present", last) ||
        need_rows(&seq_len, "bc_sequence_length", last) || need_rows(&qscore, "bc_mean_qscore", last) ||
        need_rows(&n_events, "bc_num_events", last) || need_rows(&seq_off, "seq_offsets", last + 1))
        goto done;
    /* poly(A): the spike rows of all records back to back, rows of record k = [off[k], off[k + 1]) */
    if (polya && spikes_obj != Py_None && spike_off_obj != Py_None) {
        if (PyObject_GetBuffer(spikes_obj, &spikes.view, PyBUF_C_CONTIGUOUS) < 0) goto done;
        spikes.held = 1;
        if (PyObject_GetBuffer(spike_off_obj, &spike_off.view, PyBUF_C_CONTIGUOUS) < 0) goto done;
        spike_off.held = 1;
        if (spikes.view.itemsize != 4 || spike_off.view.itemsize != 8 || spike_off.view.len / 8 < n + 1) {
            PyErr_SetString(PyExc_TypeError, "report_run: spikes must be float32 rows of four, spike offsets int64 [n + 1]");
            goto done;
        }
    }
    /* reads somebody else reports (the chimera scan found candidates in them): their slots are left None */
    if (skip_obj != Py_None) {
        if (PyObject_GetBuffer(skip_obj, &skip.view, PyBUF_C_CONTIGUOUS) < 0) goto done;
        skip.held = 1;
        if (skip.view.itemsize != 1 || skip.view.len < n) {
            PyErr_SetString(PyExc_TypeError, "report_run: skip must be a bool per read");
            goto done;
        }
    }
    /* reads too short for the scaler (the gate of load_padded_signal_head, signal_loader.py:212-222, decided on the host
     * from the read's metadata): 'scaler_signal_too_short', whatever the pass made of their samples */
    if (short_obj != Py_None) {
        if (PyObject_GetBuffer(short_obj, &too_short.view, PyBUF_C_CONTIGUOUS) < 0) goto done;
        too_short.held = 1;
        if (too_short.view.itemsize != 1 || too_short.view.len < n) {
            PyErr_SetString(PyExc_TypeError, "report_run: short must be a bool per read");
            goto done;
        }
    }
    PyObject *filenames = get_list(cols, "filenames", 0), *read_ids = get_list(cols, "read_ids", 0);
    if (!filenames || !read_ids) goto done;
    /* `seq_base`: the text arenas hold the reads of THIS call only (a run of a FAST5 file whose other columns describe the
     * whole file, fast5_file.FileRunColumns): sequence offsets count from the run's first read */
    long long seq_base = 0;
    {
        PyObject* sb = PyDict_GetItemString(cols, "seq_base");           /* borrowed */
        if (sb && sb != Py_None) {
            seq_base = PyLong_AsLongLong(sb);
            if (seq_base == -1 && PyErr_Occurred()) goto done;
        }
    }
    out = PyList_New(n);
    if (!out) goto done;
    {
        const pxg_read_result* R = (const pxg_read_result*)rec.view.buf;
        const int64_t* so = (const int64_t*)seq_off.view.buf;
        PyObject* zero = PyLong_FromLong(0);
        if (!zero) { Py_CLEAR(out); goto done; }
        for (Py_ssize_t k = 0; k < n; k++) {
            const Py_ssize_t b = first + k;
            const pxg_read_result* r = R + k;
            if (skip.held && ((const uint8_t*)skip.view.buf)[k]) {
                Py_INCREF(Py_None);
                PyList_SET_ITEM(out, k, Py_None);
                continue;
            }
            /* the rules, in the order the general path applies them (signal_loader.attach_records,
             * SignalAnalyzer.judge, BarcodeDemultiplexer.assign, SignalAnalyzer.bulk_base_space) */
            int status = PXG_ST_OKAY, label = -1, called = 0, summary = 0, tail = 0;
            if (too_short.held && ((const uint8_t*)too_short.view.buf)[k]) {
                status = PXG_ST_SCALER_SIGNAL_TOO_SHORT;             /* (:220-222: stops before the scaler, without a label) */
            } else if (r->status == PXG_ST_SCALING_QC_FAIL) {
                status = PXG_ST_SCALING_QC_FAIL;                     /* (:108-109: stops without a label) */
            } else if (r->seg_first[adapter] < 0) {
                status = PXG_ST_ADAPTER_NOT_DETECTED, label = 1;     /* 'fail' */
            } else {
                called = barcoding && r->bc_pushed && r->bc_called;
                tail = polya && r->polya_called;                     /* (both decided before anything base-space can fail) */
                if (!((const uint8_t*)present.view.buf)[b]) {
                    status = PXG_ST_NOT_BASECALLED, label = 1;
                } else {
                    summary = 1;
                    if (so[b + 1] - so[b] < min_seq_len) status = PXG_ST_SEQUENCE_TOO_SHORT, label = 1;
                    else label = 0;                                  /* 'pass' */
                }
            }
            PyObject* d = _PyDict_NewPresized(20);
            PyObject* o;
            if (!d) goto fail_row;
            if (!(o = item(filenames, b))) goto fail_row;
            SET_BORROWED(d, KEYS[K_FILENAME], o);
            if (!(o = item(read_ids, b))) goto fail_row;
            SET_BORROWED(d, KEYS[K_READ_ID], o);
            SET_BORROWED(d, KEYS[K_STATUS], PyTuple_GET_ITEM(status_names, status));
            SET(d, KEYS[K_CHANNEL], ucs4_item(&channel, b));
            {
                const double rate = ((const double*)calib.view.buf)[4 * b + 3];       /* pxg_calib.sampling_rate */
                if (rate == 0.0) { PyErr_SetString(PyExc_ZeroDivisionError, "float division by zero"); goto fail_row; }
                SET(d, KEYS[K_START_TIME], round3((double)((const int64_t*)start_time.view.buf)[b] / rate));
            }
            SET(d, KEYS[K_RUN_ID], ucs4_item(&run_id, b));
            SET(d, KEYS[K_SAMPLE_ID], ucs4_item(&sample_id, b));
            SET(d, KEYS[K_DURATION], PyLong_FromLongLong(((const int64_t*)duration.view.buf)[b]));
            if (summary) {
                SET(d, KEYS[K_NUM_EVENTS], PyLong_FromLongLong(((const int64_t*)n_events.view.buf)[b]));
                SET(d, KEYS[K_SEQUENCE_LENGTH], PyLong_FromLongLong(((const int64_t*)seq_len.view.buf)[b]));
                /* the table keeps the summary's float32 in a float64 column */
                SET(d, KEYS[K_MEAN_QSCORE], PyFloat_FromDouble((double)(float)((const double*)qscore.view.buf)[b]));
                const int64_t lo = so[b] - seq_base, hi = so[b + 1] - seq_base;
                if (lo < 0 || hi < lo || hi > seq_arena.view.len || hi > qual_arena.view.len) {
                    PyErr_SetString(PyExc_IndexError, "report_run: sequence offsets outside the arena");
                    goto fail_row;
                }
                PyObject* s = PyUnicode_DecodeASCII((const char*)seq_arena.view.buf + lo, hi - lo, NULL);
                PyObject* q = s ? PyUnicode_DecodeASCII((const char*)qual_arena.view.buf + lo, hi - lo, NULL) : NULL;
                PyObject* t = q ? PyTuple_Pack(3, s, q, zero) : NULL;
                Py_XDECREF(s); Py_XDECREF(q);
                if (t) PyObject_GC_UnTrack(t);          /* (str, str, int): see report() */
                SET(d, KEYS[K_SEQUENCE], t);
            } else {
                SET_BORROWED(d, KEYS[K_NUM_EVENTS], zero);
                SET_BORROWED(d, KEYS[K_SEQUENCE_LENGTH], zero);
                SET_BORROWED(d, KEYS[K_MEAN_QSCORE], zero);
            }
            if (label >= 0) SET_BORROWED(d, KEYS[K_LABEL], PyTuple_GET_ITEM(label_names, label));
            if (called) {
                SET(d, KEYS[K_BARCODE], PyLong_FromLong(r->bc_label));
                SET(d, KEYS[K_BARCODE_GUESS], PyLong_FromLong(r->bc_label));
                SET(d, KEYS[K_BARCODE_SCORE], PyLong_FromLong(r->bc_phred));
            }
            if (tail) {
                const double rate = ((const double*)calib.view.buf)[4 * b + 3];
                Py_ssize_t ns = 0;
                int64_t at = 0;
                if (spikes.held) {                                   /* (no spike table: tails without their spikes) */
                    const Py_ssize_t spike_rows = spikes.view.len / 16;
                    ns = r->polya_n_spikes;
                    at = ((const int64_t*)spike_off.view.buf)[k];
                    if (!spike_rows) ns = 0, at = 0;
                    if (ns < 0 || at < 0 || at + ns > spike_rows) {
                        PyErr_SetString(PyExc_IndexError, "report_run: spike rows outside the table");
                        goto fail_row;
                    }
                }
                SET(d, KEYS[K_POLYA], polya_dict(r->polya_begin, r->polya_end, (double)r->polya_dwell_samples / rate, ns,
                                                 ns ? (const float*)spikes.view.buf + (size_t)at * 4 : NULL));
            }
            PyList_SET_ITEM(out, k, d);
            continue;
        fail_row:
            Py_XDECREF(d);
            Py_CLEAR(out);
            break;
        }
        Py_DECREF(zero);
    }
done:
    for (size_t k = 0; k < sizeof(all) / sizeof(all[0]); k++)
        if (all[k]->held) PyBuffer_Release(&all[k]->view);
    return out;
}

/* ---- decode_and_run: the two native halves of a worker call over FAST5 files behind ONE release of the interpreter lock ----
 *
 * A reference-sized call (128 reads) over FAST5 files is three native calls with a little Python between them -- the
 * samples (pxg_h5_load_signals), the basecall text (pxg_h5_basecall_many), the GPU pass (pxg_process_batch_ex) -- and
 * every return from one of them queues for the interpreter lock again: from three worker threads on, the calls spend
 * more time waiting for it than working (tools/dev/host_cap.py --fast5: 700 calls/s with two threads, 500 with eight).
 * Here the three run back to back without the lock.  The functions come as ADDRESSES (this module links neither library;
 * native.py takes them from the libraries it has loaded) and are called through the prototypes of include/pxg.h; the
 * arrays come through the buffer protocol and stay exported for the duration of the call.
 *
 *   decode_and_run(load_signals, basecall_many, run_or_0, threads,
 *                  files, index, dst_start, n_samples, arena, signal_status,
 *                  seq_start, seq_len, seq_arena, qual_arena, move_start, n_moves, move_arena, basecall_status,
 *                  ctx, offsets, calib, stage_mask, extras_address, records) -> (decoded, rc)
 *
 * decoded: every read's samples and text arrived (both status arrays all zero); only then is the pass run.  rc: what
 * pxg_process_batch_ex returned, None when it was not called (run_or_0 == 0, or not decoded). */
typedef __typeof__(&pxg_h5_load_signals) load_signals_fn;
typedef __typeof__(&pxg_h5_basecall_many) basecall_many_fn;
typedef __typeof__(&pxg_process_batch_ex) run_batch_fn;

static PyObject* decode_and_run(PyObject* self, PyObject* args)
{
    unsigned long long a_signals, a_text, a_run, a_ctx, a_extras;
    int threads;
    unsigned int stage_mask;
    Py_buffer files, index, dst_start, n_samples, arena, sig_status, seq_start, seq_len, seq_arena, qual_arena, move_start,
        n_moves, move_arena, bc_status, offsets, calib, records;
    if (!PyArg_ParseTuple(args, "KKKiy*y*y*y*w*w*y*y*w*w*y*y*w*w*Ky*y*IKw*", &a_signals, &a_text, &a_run, &threads, &files, &index,
                          &dst_start, &n_samples, &arena, &sig_status, &seq_start, &seq_len, &seq_arena, &qual_arena,
                          &move_start, &n_moves, &move_arena, &bc_status, &a_ctx, &offsets, &calib, &stage_mask, &a_extras,
                          &records))
        return NULL;
    Py_buffer* all[] = { &files, &index, &dst_start, &n_samples, &arena, &sig_status, &seq_start, &seq_len, &seq_arena,
                         &qual_arena, &move_start, &n_moves, &move_arena, &bc_status, &offsets, &calib, &records };
    PyObject* out = NULL;
    const Py_ssize_t n = index.len / 8;
    int64_t need_samples = 0, need_text = 0, need_moves = 0;
    /* every array has a row per read, and the arenas hold what the layout says (the native readers trust the layout) */
    if (!a_signals || !a_text || files.len != n * (Py_ssize_t)sizeof(void*) || index.len != n * 8 || dst_start.len != n * 8 ||
        n_samples.len != n * 8 || sig_status.len != n * 4 || seq_start.len != n * 8 || seq_len.len != n * 8 ||
        move_start.len != n * 8 || n_moves.len != n * 8 || bc_status.len != n * 4 || offsets.len != (n + 1) * 8 ||
        calib.len != n * (Py_ssize_t)sizeof(pxg_calib) || records.len < n * (Py_ssize_t)sizeof(pxg_read_result) ||
        qual_arena.len != seq_arena.len || (a_run && !a_ctx)) {
        PyErr_SetString(PyExc_ValueError, "decode_and_run: the arrays do not describe one batch");
        goto done;
    }
    for (Py_ssize_t k = 0; k < n; k++) {
        const int64_t d = ((const int64_t*)dst_start.buf)[k], c = ((const int64_t*)n_samples.buf)[k];
        const int64_t s = ((const int64_t*)seq_start.buf)[k], sl = ((const int64_t*)seq_len.buf)[k];
        const int64_t m = ((const int64_t*)move_start.buf)[k], ml = ((const int64_t*)n_moves.buf)[k];
        if (d < 0 || c < 0 || s < 0 || sl < 0 || m < 0 || ml < 0) { need_samples = -1; break; }
        if (d + c > need_samples) need_samples = d + c;
        if (s + sl > need_text) need_text = s + sl;
        if (m + ml > need_moves) need_moves = m + ml;
    }
    if (need_samples < 0 || need_samples > arena.len / 2 || need_text > seq_arena.len || need_moves > move_arena.len ||
        (n && ((const int64_t*)offsets.buf)[n] > arena.len / 2)) {
        PyErr_SetString(PyExc_ValueError, "decode_and_run: a read lies outside its arena");
        goto done;
    }
    {
        int decoded = 1, rc = 0, ran = 0;
        Py_BEGIN_ALLOW_THREADS
        if (n) {
            ((load_signals_fn)(uintptr_t)a_signals)((int64_t)n, (const pxg_h5* const*)files.buf, (const int64_t*)index.buf,
                                                    (const int64_t*)dst_start.buf, (const int64_t*)n_samples.buf,
                                                    (int16_t*)arena.buf, threads, (int32_t*)sig_status.buf);
            ((basecall_many_fn)(uintptr_t)a_text)((int64_t)n, (const pxg_h5* const*)files.buf, (const int64_t*)index.buf,
                                                  (const int64_t*)seq_start.buf, (const int64_t*)seq_len.buf,
                                                  (uint8_t*)seq_arena.buf, (uint8_t*)qual_arena.buf,
                                                  (const int64_t*)move_start.buf, (const int64_t*)n_moves.buf,
                                                  (uint8_t*)move_arena.buf, threads, (int32_t*)bc_status.buf);
        }
        for (Py_ssize_t k = 0; k < n; k++)
            if (((const int32_t*)sig_status.buf)[k] || ((const int32_t*)bc_status.buf)[k]) { decoded = 0; break; }
        if (decoded && a_run) {
            rc = ((run_batch_fn)(uintptr_t)a_run)((pxg_ctx*)(uintptr_t)a_ctx, (int64_t)n, (const int16_t*)arena.buf,
                                                  (const int64_t*)offsets.buf, (const pxg_calib*)calib.buf, stage_mask,
                                                  (pxg_batch_extras*)(uintptr_t)a_extras, (pxg_read_result*)records.buf);
            ran = 1;
        }
        Py_END_ALLOW_THREADS
        out = ran ? Py_BuildValue("(Oi)", decoded ? Py_True : Py_False, rc)
                  : Py_BuildValue("(OO)", decoded ? Py_True : Py_False, Py_None);
    }
done:
    for (size_t k = 0; k < sizeof(all) / sizeof(all[0]); k++) PyBuffer_Release(all[k]);
    return out;
}

static PyMethodDef METHODS[] = {
    { "decode_and_run", decode_and_run, METH_VARARGS,
      "decode_and_run(native function addresses, the batch's arrays) -> (decoded, rc): FAST5 decode + GPU pass without the interpreter lock" },
    { "report", report, METH_VARARGS, "report(columns, rows) -> list of result dicts (signal_loader.py:165-198)" },
    { "report_run", report_run, METH_VARARGS,
      "report_run(bundle columns, first, n, records, adapter, barcoding, min_seq_len, status names, label names) -> list of result dicts" },
    { NULL, NULL, 0, NULL }
};

static struct PyModuleDef MODULE = { PyModuleDef_HEAD_INIT, "_pxgpy", "host-side result-dict builder", -1, METHODS };

PyMODINIT_FUNC PyInit__pxgpy(void)
{
    for (int k = 0; k < N_KEYS; k++) {
        KEYS[k] = PyUnicode_InternFromString(KEY_NAMES[k]);
        if (!KEYS[k]) return NULL;
    }
    return PyModule_Create(&MODULE);
}
