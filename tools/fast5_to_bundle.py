#!/usr/bin/env python
"""FAST5 files -> read bundles (.pxr.npz), once: the native reader (csrc/pxg_h5.cpp) decodes
whole files on host threads, the bundle keeps the samples as encoded bytes (1.2 bytes per
sample, decoded on the GPU) with columnar metadata and basecall summaries.  Real files feed the
GPU at a tenth of what it can take (host decompression, HDF5 structure walking: DESIGN section 5);
a bundle feeds it at 485-590 k reads/s.

    python tools/fast5_to_bundle.py <inputdir> <out.pxr.npz> [--raw] [--max-reads N]

The bundle lists the reads in the order poreplex_amd.session.enumerate_reads finds them (sorted
recursive walk), with file names relative to <inputdir>: a run from the bundle
(config['read_bundle']) writes what the run from the files writes."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poreplex_amd import fast5_file as F5  # noqa: E402
from poreplex_amd.config import default_config  # noqa: E402
from poreplex_amd.session import enumerate_reads  # noqa: E402


def convert(inputdir, out, compress=True, max_reads=None, log=print):
    reads, _ = enumerate_reads(default_config(inputdir=inputdir))
    if max_reads:
        reads = reads[:max_reads]
    files, index, names, skipped = [], [], [], []
    opened = {}
    for filename, read_id in reads:
        f = opened.get(filename)
        if f is None:
            f = opened[filename] = F5.open_fast5(os.path.join(inputdir, filename))
        i = f.index_of(read_id) if f.multi else 0
        if i < 0 or f.info['status'][i]:
            skipped.append((filename, read_id))
            continue
        files.append(f); index.append(i); names.append(filename)
    b = F5.Fast5Batch(files, index, names).as_bundle()
    bad = np.nonzero((b.signal_status != 0) | (b.basecall_status != 0))[0]
    if len(bad):
        raise SystemExit('{} read(s) cannot be decoded, first: {} #{}'.format(
            len(bad), names[bad[0]], b.read_ids[bad[0]]))
    d = b.d
    basecalls = None            # the columns go in as they are (no per-read dicts)
    cols = {k: d[k] for k in ('duration', 'start_time', 'channel_number', 'run_id', 'sample_id')}
    F5.write_bundle(out, d['arena'], d['offsets'], d['calib'], d['filename'], d['read_id'], basecalls=basecalls,
                    compress=compress, **cols)
    # write_bundle filled neutral basecall columns: replace them with the real ones
    with np.load(out, allow_pickle=False) as npz:
        full = {k: npz[k] for k in npz.files}
    for k in F5.BASECALL_COLUMNS:
        full[k] = d[k]
    np.savez(out, **full)
    log('{} reads from {} file(s) -> {} ({:.1f} MB, {} skipped)'.format(
        len(names), len(opened), out, os.path.getsize(out) / 1e6, len(skipped)))
    return len(names), skipped


if __name__ == '__main__':
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('inputdir')
    ap.add_argument('out')
    ap.add_argument('--raw', action='store_true', help='keep the samples as int16 (2 bytes per sample)')
    ap.add_argument('--max-reads', type=int, default=None)
    a = ap.parse_args()
    convert(a.inputdir, a.out, compress=not a.raw, max_reads=a.max_reads)
