#!/usr/bin/env python
"""Summarise rocprofv3 csv output (kernel stats + per-kernel PMC means)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, pattern), recursive=True))


def short(name):
    name = name.split('(')[0]
    for k in ('k_scaler_lstm_q8_lat', 'k_demux_bidir_q8_lat', 'k_demux_top_q8_lat', 'k_scaler_lstm_q8', 'k_demux_bidir_q8', 'k_demux_top_q8', 'k_scaler_lstm_q', 'k_scaler_lstm', 'k_demux_bidir', 'k_demux_top', 'k_viterbi_ltr', 'k_head_pool',
              'k_barcode_window_raw', 'k_finalize', 'k_compact_ok', 'k_scaler_transform',
              'k_mark_pushed', 'k_polya_order_count', 'k_polya_order_starts', 'k_polya_order_place', 'k_polya', 'k_events', 'k_guppy_event_means', 'k_unsplit_scan_w', 'k_unsplit_scan',
              'k_pool_scale', 'k_raw_to_pa', 'k_barcode_window_f32', 'k_detect_events'):
        if k in name:
            return k
    return name[:40]


for f in find('trace/**/*kernel_stats.csv'):
    print('== kernel stats', f)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print('  {:28s} calls {:>5s} total_ns {:>12s} avg_ns {:>12s} pct {:>6s}'.format(
                short(row['Name']), row['Calls'], row['TotalDurationNs'],
                row['AverageNs'], row['Percentage']))
for d in sorted(glob.glob(os.path.join(root, 'pmc*'))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, '**/*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    print('== PMC', d)
    for k, cs in sorted(acc.items()):
        print('  ' + k)
        for c, vals in sorted(cs.items()):
            print('      {:28s} mean/launch {:16.1f}  (n={})'.format(c, sum(vals) / len(vals), len(vals)))
