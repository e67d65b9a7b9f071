#!/usr/bin/env python
"""<tag>_summary.txt (tools/prof_summary.py output of a tools/prof.sh run) -> per-kernel bound figures as JSON
(profiles/<round>/full_kernel_bounds.json; bench.py quotes it in extra.full / extra.kernel_bounds).
usage: tools/prof_bounds.py <summary.txt> [kernel ...] > full_kernel_bounds.json
Derivations (all counters are means per launch):
  clock_GHz                       GRBM_GUI_ACTIVE / 8 XCDs / average duration
  valu_issue_frac_of_simd_cycles  SQ_ACTIVE_INST_VALU x 4 (quad-cycles -> cycles) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
  mfma_busy_frac_of_simd_cycles   SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
  wait_any / wait_inst            SQ_WAIT_ANY, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  salu_share_of_instructions      SQ_INSTS_SALU / (SQ_INSTS_VALU + SQ_INSTS_SALU + SQ_INSTS_SMEM + SQ_INSTS_LDS)
  lds_bank_conflict_frac          SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  lds_busy_frac                   SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8)
  hbm_bytes                       FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes"""
import json
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:] or ['k_polya', 'k_unsplit_scan_w', 'k_unsplit_scan', 'k_guppy_event_means', 'k_viterbi_ltr', 'k_scaler_lstm_q8',
                        'k_demux_bidir_q8', 'k_demux_top_q8']
avg = {}
for m in re.finditer(r'^\s+(\S+)\s+calls\s+(\d+)\s+total_ns\s+\d+\s+avg_ns\s+([\d.]+)', txt, re.M):
    avg[m.group(1)] = (float(m.group(3)), int(m.group(2)))
cnt = {}
for sec in re.finditer(r'== PMC \S+\n(.*?)(?=\n== |\Z)', txt, re.S):
    cur = None
    for line in sec.group(1).splitlines():
        k = re.match(r'^  (\S+)$', line)
        if k:
            cur = k.group(1)
            continue
        v = re.match(r'^\s+(\S+)\s+mean/launch\s+([\d.]+)', line)
        if v and cur:
            cnt.setdefault(cur, {})[v.group(1)] = float(v.group(2))
out = {'_note': __doc__.split('Derivations')[1].strip(), '_source': sys.argv[1], 'kernels': {}}
for k in want:
    name = k if k in avg else next((n for n in avg if n.startswith(k + '<') or n.startswith(k + '(')), None)
    c = cnt[k] if k in cnt else next((cnt[n] for n in cnt if n.startswith(k + '<') or n.startswith(k + '(')), None)
    if name is None or c is None:
        continue
    ns, calls = avg[name]
    cyc = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    row = {'avg_ms': round(ns / 1e6, 4), 'launches': calls}
    if cyc:
        row['clock_GHz'] = round(cyc / ns, 4)
        row['valu_issue_frac_of_simd_cycles'] = round(c.get('SQ_ACTIVE_INST_VALU', 0) * 4 / (1024 * cyc), 4)
        row['mfma_busy_frac_of_simd_cycles'] = round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc), 4)
        row['lds_busy_frac'] = round(c.get('SQ_LDS_IDX_ACTIVE', 0) / (256 * cyc), 4)
    wc = c.get('SQ_WAVE_CYCLES', 0)
    if wc:
        row['wait_any_frac_of_wave_cycles'] = round(c.get('SQ_WAIT_ANY', 0) / wc, 4)
        row['wait_inst_frac_of_wave_cycles'] = round(c.get('SQ_WAIT_INST_ANY', 0) / wc, 4)
    tot = sum(c.get(x, 0) for x in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INSTS_LDS'))
    if tot:
        row['salu_share_of_instructions'] = round(c.get('SQ_INSTS_SALU', 0) / tot, 4)
    if c.get('SQ_LDS_IDX_ACTIVE'):
        row['lds_bank_conflict_frac_of_lds_cycles'] = round(c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE'], 4)
    if 'FETCH_SIZE' in c or 'WRITE_SIZE' in c:
        row['hbm_bytes'] = (2 * c.get('FETCH_SIZE', 0) + c.get('WRITE_SIZE', 0)) * 1024
        row['hbm_GBps'] = round(row['hbm_bytes'] / ns, 4)
    out['kernels'][k] = row
print(json.dumps(out, indent=1))
