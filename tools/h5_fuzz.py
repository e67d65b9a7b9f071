#!/usr/bin/env python
"""Fuzz of the native FAST5 / HDF5 reader (csrc/pxg_h5.cpp) under AddressSanitizer + UBSan: byte
flips, truncations and extreme 8-byte words in real files must end as errors or wrong samples,
never as a stray access.  Round 3 ran 15 000 mutated files clean after it found (and this fixed)
an Events-table column read past its row, field reads running off the end of a truncated map, and a
Fastq record with non-ASCII bytes reaching the text columns.

    make -C poreplex_amd/csrc asan            # -> poreplex_amd/csrc/_obj/libpxghost_asan.so
    python tests/py39/... (or tests/test_fast5_native.py) leaves sample files; any *.fast5 will do:
    LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
      ASAN_OPTIONS=detect_leaks=0 python tools/h5_fuzz.py <seed> <trials> file1.fast5 [file2.fast5 ...]
"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poreplex_amd import native as N
ASAN_LIB = os.path.join(os.path.dirname(N.TEXT_LIB_PATH), '_obj', 'libpxghost_asan.so')
if os.path.isfile(ASAN_LIB):
    N._text_lib = N.load_text_library(ASAN_LIB)
from poreplex_amd import fast5_file as F5
import glob
srcs = sys.argv[3:]
assert srcs, __doc__
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
n_open=n_fail=0
for trial in range(int(sys.argv[2]) if len(sys.argv)>2 else 1500):
    blob = bytearray(open(srcs[trial % len(srcs)],'rb').read())
    kind = trial % 3
    if kind == 0:
        for pos in rng.integers(0, len(blob), rng.integers(1, 60)): blob[pos] = rng.integers(0,256)
    elif kind == 1:
        blob = blob[:rng.integers(8, len(blob))]
    else:   # overwrite 8-byte words with extreme values (addresses / sizes)
        for pos in rng.integers(8, len(blob)-8, rng.integers(1, 12)):
            blob[pos:pos+8] = rng.choice([b'\xff'*8, b'\x00'*8, (2**63-1).to_bytes(8,'little'), int(rng.integers(0,2**40)).to_bytes(8,'little')])
    p = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'pxg_h5_fuzz.fast5'); open(p,'wb').write(bytes(blob))
    try:
        many = F5.OpenedFiles([p, p + '.absent', p], threads=2)      # the batch open: same outcome per file, as data
        assert many.rc[1] != 0 and many.rc[0] == many.rc[2]
        if many.rc[0] == 0 and many.n_reads[0] == 1 and not many.multi[0] and many.info['status'][0] == 0:
            F5.Fast5Batch.from_opened(many, [0, 2], ['a', 'b'], ['x', 'y']).as_bundle(threads=2)
        many.close()
        f = F5.Fast5File(p); n_open+=1
        info = f.info
        ok = np.nonzero(info['status']==0)[0]
        ns = np.clip(info['n_samples'][ok], 0, 1<<22)
        arena = np.zeros(int(ns.sum())+1, np.int16)
        F5.load_signals([f]*len(ok), ok, ns, arena, np.concatenate([[0],np.cumsum(ns)[:-1]]).astype(np.int64), threads=2)
        for i in ok[:3]:
            try: f.basecall(int(i))
            except OSError: pass
        if len(ok):      # the batch decoder: basecall text by the byte ranges the metadata pass noted
            F5.Fast5Batch([f] * len(ok), ok, ['x'] * len(ok)).as_bundle(threads=2)
    except OSError:
        n_fail+=1
print('trials done; opened', n_open, 'refused', n_fail)
