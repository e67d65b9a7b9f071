#!/usr/bin/env python
"""profiles/<round>/<tag>_summary.txt (tools/prof_summary.py output) -> per-kernel HBM
bytes per launch (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950,
WRITE_SIZE as reported), written next to it as <tag>_hbm_traffic.json.
usage: tools/prof_traffic.py profiles/r01/f_final_summary.txt"""
import json
import re
import sys

src = sys.argv[1]
txt = open(src).read()


def grab(section, counter):
    m = re.search(r'== PMC \S*/' + section + r'\n(.*?)(?=\n== PMC|\Z)', txt, re.S)
    out, cur = {}, None
    for line in (m.group(1) if m else '').splitlines():
        k = re.match(r'^  (\S+)$', line)
        if k:
            cur = k.group(1)
            continue
        v = re.match(r'^\s+(\S+)\s+mean/launch\s+([\d.]+)', line)
        if v and v.group(1) == counter:
            out[cur] = float(v.group(2))
    return out


f, w = grab('pmc3', 'FETCH_SIZE'), grab('pmc4', 'WRITE_SIZE')
res = {'_note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/prof.sh), mean per '
                'launch, bench.py default workload (10000 reads x ~60000 samples). Counter unit = KiB. '
                'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced '
                'reads); WRITE_SIZE as reported (uncalibrated per the guide).',
       'source': src, 'kernels': {}}
for k in sorted(f):
    if not k.startswith('__'):
        res['kernels'][k] = {'fetch_bytes': f[k] * 2048, 'write_bytes': w.get(k, 0) * 1024,
                             'hbm_bytes': f[k] * 2048 + w.get(k, 0) * 1024}
dst = src.replace('_summary.txt', '_hbm_traffic.json')
json.dump(res, open(dst, 'w'), indent=1)
print(dst, {k: round(v['hbm_bytes'] / 1e6, 1) for k, v in res['kernels'].items()})
