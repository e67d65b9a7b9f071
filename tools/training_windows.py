#!/usr/bin/env python
"""Training inputs for the barcode classifier from a run's own reads: every read goes through the GPU
path up to the classifier's input -- DAQ -> pA, learned scaling, segmentation, the adapter window cut /
padded to 300 and normalised (a9-a11) -- and the windows of the reads that were pushed are written
with their ids, the current model's call and score.  That is what the reference's training pipeline
prepares from --dump-adapter-signals (training/barcodes/scripts/prepare_training_data.py:63-87, "input
prep identical to inference"); here the inference path itself writes it.  Labels come from the user
(read id -> class, e.g. from alignments), as in the reference.

usage (GPU box): python tools/training_windows.py (--bundle reads.pxr.npz | --inputdir fast5_dir) --out windows.npz
                 [--batch-reads 10000] [--labels labels.tsv]      # read_id <TAB> class (0 = decoy, 1.. = barcodes)
Then: poreplex_amd.training.Trainer / export_demux_bundle (tests/test_training.py)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poreplex_amd import native as N  # noqa: E402
from poreplex_amd.config import default_config  # noqa: E402
from poreplex_amd.session import enumerate_reads  # noqa: E402
from poreplex_amd.signal_analyzer import SignalAnalyzer  # noqa: E402
from poreplex_amd.signal_loader import ReadTable  # noqa: E402


def collect(cfg, batch_reads=10000, log=None):
    """(windows [m, 300] float32, read ids, guess label, score) of every pushed read of the run."""
    wins, ids, guess, score = [], [], [], []
    with SignalAnalyzer(cfg, 0) as an:
        loader, ctx = an.loader, an.ctx
        loader.stage_mask = N.STAGE_ALL_DEMUX
        reads, _ = enumerate_reads(cfg, loader.bundle)
        for a in range(0, len(reads), batch_reads):
            table = ReadTable()
            an.prepare(reads[a:a + batch_reads], table)
            rows, arena, offsets, calib = loader.pack(table)
            if not len(rows):
                continue
            if isinstance(arena, N.EncodedSamples):
                ctx.stage_z(arena, offsets, calib)
            else:
                ctx.stage(arena, offsets, calib)
            ctx.swap()
            ctx.run(N.STAGE_ALL_DEMUX)
            rec = ctx.download()
            win = ctx.download_windows(rec)
            keep = rec['bc_pushed'] != 0
            wins.append(win[keep])
            ids += [table.read_id[r] for r in rows[keep].tolist()]
            guess.append(rec['bc_label'][keep].astype(np.int64))
            score.append(rec['bc_score'][keep].astype(np.float32))
            if log:
                log('{} / {} reads, {} windows'.format(min(a + batch_reads, len(reads)), len(reads), sum(map(len, wins))))
    trim = int(cfg['demultiplexing']['signal_trim_length'])
    cat = (lambda parts, dt, shape: np.concatenate(parts) if parts else np.zeros(shape, dt))
    return cat(wins, np.float32, (0, trim)), np.asarray(ids), cat(guess, np.int64, (0,)), cat(score, np.float32, (0,))


def main():
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('--bundle')
    ap.add_argument('--inputdir')
    ap.add_argument('--out', required=True)
    ap.add_argument('--batch-reads', type=int, default=10000)
    ap.add_argument('--labels')
    args = ap.parse_args()
    if bool(args.bundle) == bool(args.inputdir):
        ap.error('one of --bundle / --inputdir')
    cfg = default_config(inputdir=args.inputdir or os.path.dirname(os.path.abspath(args.bundle)),
                         outputdir=os.path.dirname(os.path.abspath(args.out)), read_bundle=args.bundle, barcoding=True)
    windows, ids, guess, score = collect(cfg, args.batch_reads, log=lambda m: print(m, file=sys.stderr))
    cols = {'windows': windows, 'read_id': ids, 'barcode_guess': guess, 'barcode_score': score}
    if args.labels:
        known = {}
        with open(args.labels) as fh:
            for line in fh:
                if line.strip():
                    rid, cls = line.split()[:2]
                    known[rid] = int(cls)
        cols['label'] = np.array([known.get(r, -1) for r in ids.tolist()], dtype=np.int64)     # -1: no label given
    np.savez_compressed(args.out, **cols)
    print('{} windows of {} samples -> {}'.format(len(windows), windows.shape[1], args.out))


if __name__ == '__main__':
    main()
