"""TEST DOUBLE of poreplex_amd.native.NativeContext, answering from oracle/libpxo.so.

Test infrastructure only: it lets the CPU suite drive the HOST logic that sits on top of the
GPU context (facade ordering / status rules, the session driver, bench.py's multi-rank
plumbing over gloo) in a container without a GPU.  Nothing under poreplex_amd/ imports it;
tests inject it (monkeypatch, or tests/bench_standin.py around bench.py's main(), whose JSON line is
then marked TEST-STANDIN and carries no value).
"""
import numpy as np

from poreplex_amd import native as N


class OracleBackedContext:
    """Same methods the host code uses, answers computed by the oracle."""

    def __init__(self, config, device_id=0):
        from oracle.pxo import Oracle
        self.oracle = Oracle(config)
        self.ncfg = self.oracle.ncfg
        self.cfg = self.oracle.cfg
        self.state_names = self.oracle.state_names
        self.n_resident = 0
        self.staged = None

    def device_info(self):
        return {'name': 'oracle test double (CPU)', 'arch': 'none', 'compute_units': 0,
                'wavefront_size': 0, 'total_mem': 0, 'lds_per_cu': 0, 'clock_khz': 0}

    def upload(self, arena, offsets, calib, scale_shift=None):
        self.batch = (np.asarray(arena), np.asarray(offsets), np.asarray(calib), scale_shift)
        self.n_resident = len(offsets) - 1

    def upload_tiled(self, n_reads, arena, offsets, calib, scale_shift=None, phase=0):
        k = len(offsets) - 1
        which = (phase + np.arange(n_reads)) % k
        parts = [arena[offsets[b]:offsets[b + 1]] for b in which]
        a, o = N.pack_reads(parts)
        self.upload(a, o, np.asarray(calib)[which],
                    None if scale_shift is None else np.asarray(scale_shift)[which])

    def stage(self, arena, offsets, calib, scale_shift=None):
        self.staged = (np.array(arena), np.array(offsets), np.array(calib), scale_shift)

    def stage_z(self, enc, offsets, calib, scale_shift=None):
        self.stage(enc.decode(), offsets, calib, scale_shift)       # the host reference decoder

    def swap(self):
        self.upload(*self.staged)
        self.staged = None

    def pin(self, array):
        return array

    def unpin(self, array):
        pass

    def run(self, mask=N.STAGE_ALL_DEMUX):
        self.res, self.spk = self.oracle.process_batch(*self.batch, stage_mask=mask,
                                                       want_spikes=True)

    def sync(self):
        pass

    def download(self, out=None):
        if out is not None:
            out[:len(self.res)] = self.res
            return out[:len(self.res)]
        return self.res

    def download_spikes(self, records=None):
        from oracle.pxo import spikes_csr
        return spikes_csr(self.res, self.spk)

    def stage_times(self):
        return {k: 0.0 for k in N.TIMER_NAMES}, {k: 0 for k in N.TIMER_NAMES}

    def unsplit_scan(self, first_sample, n_blocks, block_stride=15):
        arena, offsets, calib, _ = self.batch
        n = len(offsets) - 1
        found = []
        cnt = np.zeros(n, dtype=np.int32)
        a = int(self.cfg.segmentation_model.adapter_state)
        for i in range(n):
            r = self.res[i]
            if n_blocks[i] <= 0 or r['status'] != 0 or r['seg_first'][a] < 0:
                continue
            _, scaled = self.oracle.guppy_event_means(
                arena[offsets[i]:offsets[i + 1]], calib[i], first_sample[i], n_blocks[i],
                r['scale'], r['shift'], block_stride)
            got, c = self.oracle.unsplit_scan(scaled, first_sample[i],
                                              (int(r['seg_last'][a]) + 1) * int(self.cfg.stride),
                                              float(calib[i]['sampling_rate']), block_stride)
            found.append(got)
            cnt[i] = c
        start = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        iv = np.concatenate(found) if found else np.zeros((0, 2), dtype=np.int64)
        return iv.reshape(-1, 2), cnt, start

    def unsplit_scan_events(self, n_events, ev_start, ev_mean):
        arena, offsets, calib, _ = self.batch
        n = len(offsets) - 1
        found = []
        cnt = np.zeros(n, dtype=np.int32)
        a = int(self.cfg.segmentation_model.adapter_state)
        eo = np.concatenate([[0], np.cumsum(np.maximum(np.asarray(n_events, dtype=np.int64), 0))]).astype(np.int64)
        for i in range(n):
            r = self.res[i]
            if n_events[i] <= 0 or r['status'] != 0 or r['seg_first'][a] < 0:
                continue
            got, c = self.oracle.unsplit_scan_events(ev_mean[eo[i]:eo[i + 1]], ev_start[eo[i]:eo[i + 1]], r['scale'],
                                                     r['shift'], (int(r['seg_last'][a]) + 1) * int(self.cfg.stride),
                                                     float(calib[i]['sampling_rate']))
            found.append(got)
            cnt[i] = c
        start = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        iv = np.concatenate(found) if found else np.zeros((0, 2), dtype=np.int64)
        return iv.reshape(-1, 2), cnt, start

    def event_table(self, first_sample, n_blocks, block_stride=15):
        arena, offsets, calib, _ = self.batch
        n = len(offsets) - 1
        off = np.concatenate([[0], np.cumsum(np.maximum(np.asarray(n_blocks, dtype=np.int64), 0))]).astype(np.int64)
        mean, stdv, scaled = (np.zeros(int(off[-1]), dtype=np.float32) for _ in range(3))
        for i in np.nonzero(np.diff(off))[0].tolist():
            r = self.res[i]
            mean[off[i]:off[i + 1]], stdv[off[i]:off[i + 1]], scaled[off[i]:off[i + 1]] = self.oracle.guppy_event_table(
                arena[offsets[i]:offsets[i + 1]], calib[i], first_sample[i], n_blocks[i], r['scale'], r['shift'],
                block_stride)
        return mean, stdv, scaled, off

    def download_windows(self, records=None):
        arena, offsets, calib, _ = self.batch
        a = int(self.cfg.segmentation_model.adapter_state)
        out = np.zeros((len(offsets) - 1, int(self.cfg.signal_trim_length)), dtype=np.float32)
        for i in np.nonzero(self.res['bc_pushed'])[0].tolist():
            r = self.res[i]
            sig = self.oracle.pool_scale(arena[offsets[i]:offsets[i + 1]], calib[i], r['scale'], r['shift'])
            out[i] = self.oracle.barcode_window(sig[int(r['seg_first'][a]):int(r['seg_last'][a]) + 1])[0]
        return out

    def pooled_signal(self, first, count):
        arena, offsets, calib, _ = self.batch
        n = len(offsets) - 1
        out_off = np.concatenate([[0], np.cumsum(np.maximum(np.asarray(count, dtype=np.int64), 0))]).astype(np.int64)
        out = np.zeros(int(out_off[-1]), dtype=np.float32)
        for i in np.nonzero(np.diff(out_off))[0].tolist():
            r = self.res[i]
            sig = self.oracle.pool_scale(arena[offsets[i]:offsets[i + 1]], calib[i], r['scale'], r['shift'])
            out[out_off[i]:out_off[i + 1]] = sig[int(first[i]):int(first[i]) + int(count[i])]
        return out, out_off

    def close(self):
        pass


class OneCallOracleContext(OracleBackedContext):
    """The double with the ONE-CALL form of a worker batch (native.NativeContext.process_batch_ex: stage, swap, run,
    downloads and the window scan behind one call), so that the CPU suite drives the host code the way the GPU does:
    SignalLoader.fit_scalers' first branch and SignalAnalyzer.process_plain_run with the chimera scan on."""

    calls = 0

    def process_batch_ex(self, samples, offsets, calib, stage_mask=N.STAGE_ALL_DEMUX, scale_shift=None, unsplit=None,
                         want_spikes=False):
        import threading
        lock = self.__dict__.setdefault('_one_call', threading.Lock())
        with lock:                                    # (the double keeps ONE resident batch)
            type(self).calls += 1
            if isinstance(samples, N.EncodedSamples):
                samples = samples.decode()
            self.upload(np.array(samples, dtype=np.int16), np.array(offsets, dtype=np.int64), np.array(calib), scale_shift)
            self.run(stage_mask)
            out = {'records': np.array(self.download())}
            if want_spikes:
                out['spikes'] = self.download_spikes(out['records'])
            if unsplit is not None:
                out['unsplit'] = self.unsplit_scan(np.asarray(unsplit[0]), np.asarray(unsplit[1]), int(unsplit[2]))
            return out


class NativeEntryMixin:
    """Gives a double that has process_batch_ex the NATIVE face of that call: a C function pointer with
    pxg_process_batch_ex's prototype (a ctypes callback) behind `lib.pxg_process_batch_ex`, and a `handle`, so that
    native.BatchExCall and csrc/pxg_pyreport.c decode_and_run drive it exactly as they drive the library: raw pointers
    in, records / spike rows / candidate intervals written through the extras struct, PXG_E_NOMEM with the totals set
    when a buffer is too small."""

    PROTOTYPE = None
    native_calls = 0

    def install_native_entry(self):
        import ctypes as C
        import types
        cls = NativeEntryMixin
        if cls.PROTOTYPE is None:
            cls.PROTOTYPE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                        C.c_void_p, C.c_void_p)
        self._entry = cls.PROTOTYPE(self._native_entry)            # (kept alive with the double)
        self.lib = types.SimpleNamespace(pxg_process_batch_ex=self._entry)
        self.handle = C.c_void_p(0x5eed)

    def _check(self, rc, what):
        if rc != 0:
            raise N.PxgError('{} failed ({})'.format(what, rc))

    def batch_ex_call(self, n, stage_mask=N.STAGE_ALL_DEMUX, unsplit=None, want_spikes=False):
        if not hasattr(self, '_entry'):
            self.install_native_entry()
        return N.BatchExCall(self, n, stage_mask, unsplit, want_spikes)

    def _native_entry(self, handle, n, arena_p, offsets_p, calib_p, mask, extras_p, out_p):
        import ctypes as C
        try:
            assert handle == 0x5eed

            def view(address, count, dtype):
                dtype = np.dtype(dtype)
                if not count:
                    return np.zeros(0, dtype=dtype)
                return np.frombuffer((C.c_char * (count * dtype.itemsize)).from_address(address), dtype=dtype)
            offsets = view(offsets_p, n + 1, np.int64).copy()
            arena = view(arena_p, int(offsets[-1]), np.int16)
            calib = view(calib_p, n, N.CALIB_DTYPE)
            x = N.PxgBatchExtras.from_address(extras_p)
            assert x.struct_bytes == C.sizeof(N.PxgBatchExtras) and not x.z and not x.scale_shift_or_null
            unsplit = None
            if x.unsplit_first_sample:
                unsplit = (view(x.unsplit_first_sample, n, np.int64), view(x.unsplit_n_blocks, n, np.int64),
                           int(x.unsplit_block_stride))
            res = self.process_batch_ex(arena, offsets, calib, mask, unsplit=unsplit, want_spikes=bool(x.spike_cap))
            view(out_p, n, N.RESULT_DTYPE)[:] = res['records']
            rc = 0
            if x.spike_cap:
                rows, off = res['spikes']
                x.spike_total = len(rows)
                view(x.spike_offsets, n + 1, np.int64)[:] = off
                if len(rows) > x.spike_cap:
                    rc = N.PXG_E_NOMEM
                else:
                    view(x.spikes, 4 * len(rows), np.float32)[:] = np.asarray(rows, dtype=np.float32).ravel()
            if unsplit is not None:
                iv, cnt, _ = res['unsplit']
                x.unsplit_total = len(iv)
                view(x.unsplit_count, n, np.int32)[:] = cnt
                if len(iv) > x.unsplit_cap:
                    rc = N.PXG_E_NOMEM
                else:
                    view(x.unsplit_intervals, 2 * len(iv), np.int64)[:] = np.asarray(iv, dtype=np.int64).ravel()
            type(self).native_calls += 1
            return rc
        except BaseException:             # noqa: BLE001  (nothing may leave a C callback)
            import traceback
            traceback.print_exc()
            return -1


class NativeOneCallOracleContext(NativeEntryMixin, OneCallOracleContext):
    """OneCallOracleContext that can also be called the way the library is (see NativeEntryMixin)."""

