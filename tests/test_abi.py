"""The C-ABI library loads and exports every symbol include/pxg.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

from poreplex_amd import native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'pxg.h')).read()
    return sorted(set(re.findall(r'^\s*(?:int|int64_t|void|const char\*)\s+(pxg_\w+)\s*\(', text, re.M)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(N.EXPORTED_SYMBOLS + N.TEXT_SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.isfile(N.LIB_PATH):
        pytest.skip('libpxg.so not built (run __graft_entry__.build())')
    lib = ctypes.CDLL(N.LIB_PATH)
    text = ctypes.CDLL(N.TEXT_LIB_PATH)          # the host-only sink text lives in its own library
    for name in declared_symbols():
        getattr(text if name in N.TEXT_SYMBOLS else lib, name)
    lib.pxg_abi_version.restype = ctypes.c_int
    assert lib.pxg_abi_version() == N.PXG_ABI_VERSION


def test_struct_layouts_match_the_header_sizes():
    # sizes the C compiler gives the ABI structs (computed from the header by gcc)
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "pxg.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu", sizeof(pxg_config), sizeof(pxg_read_result), sizeof(pxg_hmm), sizeof(pxg_event), sizeof(pxg_calib), sizeof(pxg_stage_times), sizeof(pxg_summary_columns), sizeof(pxg_batch_extras), sizeof(pxg_z_chunk));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 'sz.c')
        open(c, 'w').write(src)
        exe = os.path.join(d, 'sz')
        subprocess.check_call(['gcc', '-I' + os.path.join(ROOT, 'include'), c, '-o', exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    want = [ctypes.sizeof(N.PxgConfig), ctypes.sizeof(N.PxgReadResult), ctypes.sizeof(N.PxgHmm),
            ctypes.sizeof(N.PxgEvent), ctypes.sizeof(N.PxgCalib), ctypes.sizeof(N.PxgStageTimes),
            ctypes.sizeof(N.PxgSummaryColumns), ctypes.sizeof(N.PxgBatchExtras), N.Z_CHUNK_DTYPE.itemsize]
    assert sizes == want


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(N, '_lib', None)
    with pytest.raises(N.PxgError, match='no CPU fallback'):
        N.load_library(str(tmp_path / 'absent.so'))


def test_page_exclusive_arrays_own_their_pages():
    """native.page_exclusive / pinnable: what gets page-locked shares no page with another allocation (hipHostRegister
    works on pages)."""
    import numpy as np
    from poreplex_amd import native as N
    import mmap
    for n, dt in ((1, np.int16), (1000, N.RESULT_DTYPE), (4096, np.uint8), (100001, np.int64)):
        a = N.page_exclusive(n, dt, fill=0)
        assert a.shape == (n,) and a.ctypes.data % 4096 == 0 and not a.view(np.uint8).any()
        # cut from an anonymous mapping of its own (never the malloc heap: a heap range that was page-locked once
        # must not meet the runtime's in-place lock of a pageable copy source again)
        base = a
        while isinstance(base, np.ndarray) and base.base is not None:
            base = base.base
        assert isinstance(getattr(base, 'obj', base), mmap.mmap), type(base)
        a.view(np.uint8)[...] = 1                     # writable
    small = np.arange(1000, dtype=np.int16)
    p = N.pinnable(small)
    assert p is not small and np.array_equal(p, small) and p.ctypes.data % 4096 == 0
    chunks = np.zeros(7, dtype=N.Z_CHUNK_DTYPE)
    chunks['len'] = 5
    assert np.array_equal(N.pinnable(chunks), chunks)
