#!/usr/bin/env python
"""Self-check against accidental copying: longest runs of identical (whitespace-normalised,
non-blank) lines between a file here and its namesake in the reference tree, plus the
share of this file's code lines that occur verbatim anywhere in that reference file.
Runs only where /root/reference exists (development container)."""
import os
import re
import sys

PAIRS = [('poreplex_amd/signal_analyzer.py', 'poreplex/signal_analyzer.py'),
         ('poreplex_amd/signal_loader.py', 'poreplex/signal_loader.py'),
         ('poreplex_amd/barcoding.py', 'poreplex/barcoding.py'),
         ('poreplex_amd/polya.py', 'poreplex/polya.py'),
         ('poreplex_amd/fast5_file.py', 'poreplex/fast5_file.py'),
         ('poreplex_amd/sinks.py', 'poreplex/io.py'),
         ('poreplex_amd/session.py', 'poreplex/pipeline.py'),
         ('poreplex_amd/utils.py', 'poreplex/utils.py'),
         ('poreplex_amd/worker_persistence.py', 'poreplex/worker_persistence.py')]


def norm(path):
    out = []
    for no, line in enumerate(open(path, errors='replace'), 1):
        t = re.sub(r'\s+', ' ', line.strip())
        if t and not t.startswith('#'):
            out.append((no, t))
    return out


def main(ref_root='/root/reference', min_block=4):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worst = 0
    for mine, theirs in PAIRS:
        a, b = os.path.join(here, mine), os.path.join(ref_root, theirs)
        if not (os.path.isfile(a) and os.path.isfile(b)):
            continue
        A, B = norm(a), norm(b)
        bset = set(t for _, t in B)
        shared = sum(1 for _, t in A if t in bset and len(t) > 12)
        blocks = []
        i = 0
        while i < len(A):
            best = 0
            for j in range(len(B)):
                k = 0
                while i + k < len(A) and j + k < len(B) and A[i + k][1] == B[j + k][1]:
                    k += 1
                best = max(best, k)
            if best >= min_block:
                blocks.append((A[i][0], best))
                i += best
            else:
                i += 1
        worst = max(worst, max([n for _, n in blocks], default=0))
        print('{:40s} code lines {:4d}  verbatim-in-ref {:4d} ({:.0%})  identical blocks >= {}: {}'.format(
            mine, len(A), shared, shared / max(len(A), 1), min_block, blocks))
    return worst


if __name__ == '__main__':
    sys.exit(1 if main() >= 4 else 0)
