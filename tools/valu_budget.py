#!/usr/bin/env python
"""Instruction budget of the step loop of the LSTM kernels, from the ISA hipcc emits ("treat the LSTM
kernels' VALU as the budget it is").  usage: tools/valu_budget.py > profiles/r06/k2_valu_budget.txt
The step loop is found as the innermost loop that contains the kernel's v_mfma instructions; opcodes are counted per
class and the VALU are attributed to the groups of the arithmetic (the attribution is by opcode and by the known
shape of each group -- a look-up is v_med3 + v_fract + v_cvt_flr + v_lshl_add (address) + 3 fma, ...)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'poreplex_amd', 'csrc')
FLAGS = '--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -S --cuda-device-only'.split()


def isa(src, extra=()):
    out = os.path.join(tempfile.mkdtemp(prefix='pxg_isa_'), 'k.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + FLAGS + list(extra) + ['-o', out, os.path.join(CSRC, src)],
                          stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def kernel_body(lines, name):
    start = next(i for i, ln in enumerate(lines) if re.match(r'^_Z\d+' + name + r'\w*:', ln))
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    return lines[start:end + 1]


def step_loop(body):
    """every instruction of the innermost loop that holds the kernel's v_mfma instructions, by LLVM's own block
    annotations ("in Loop: Header=BBn_m", "This Inner Loop Header"): a rotated loop's blocks lie on both sides of its
    header in the listing and all of them belong to a steady-state step."""
    blocks, cur, header = [], None, None
    for ln in body:
        m = re.match(r'^(?:\.L(BB\d+_\d+):|; %bb\.\d+:)(.*)$', ln)
        if m:
            note = m.group(2)
            h = re.search(r'Header=(BB\d+_\d+)', note)
            if 'Loop Header' in note and m.group(1):
                header = m.group(1) if 'Inner Loop Header' in note else header
                hdr = m.group(1)
            else:
                hdr = h.group(1) if h else None
            cur = {'header': hdr, 'lines': []}
            blocks.append(cur)
        elif cur is not None:
            # the comment lines that follow a label carry the rest of its annotation
            if ln.strip().startswith(';') and not cur['lines']:
                h = re.search(r'Header=(BB\d+_\d+)', ln)
                if h and cur['header'] is None:
                    cur['header'] = h.group(1)
                continue
            cur['lines'].append(ln)
    owners = collections.Counter(b['header'] for b in blocks if b['header'] and any('v_mfma' in x for x in b['lines']))
    owner = owners.most_common(1)[0][0]         # (peeled first / last iterations hold MFMAs outside any loop)
    return [x for b in blocks if b['header'] == owner for x in b['lines']]


def classify(op):
    if op.startswith('v_mfma'):
        return 'MFMA'
    if op.startswith('v_'):
        return 'VALU'
    if op.startswith('ds_'):
        return 'LDS'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'VMEM'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'):
        return 'wait/nop'
    if op.startswith('s_'):
        return 'SALU'
    return 'other'


def report(title, src, kname, groups, extra=()):
    loop = step_loop(kernel_body(isa(src, extra), kname))
    ops = collections.Counter()
    for ln in loop:
        t = ln.strip()
        if not t or t.startswith((';', '.')) or t.endswith(':'):
            continue
        ops[t.split()[0]] += 1
    cls = collections.Counter()
    for op, c in ops.items():
        cls[classify(op)] += c
    print('=' * 100)
    print(title)
    print('  step loop: {} instructions: '.format(sum(ops.values())) + ', '.join('{} {}'.format(k, v) for k, v in cls.most_common()))
    print('  VALU by opcode: ' + ', '.join('{} {}'.format(k, v) for k, v in sorted(ops.items(), key=lambda kv: -kv[1]) if classify(k) == 'VALU'))
    left = cls['VALU']
    print('  VALU by group of the arithmetic:')
    for name, count, how in groups(ops):
        print('    {:44s} {:4d}   {}'.format(name, count, how))
        left -= count
    print('    {:44s} {:4d}   (addresses, loop control in vector registers, x tile, moves)'.format('everything else', left))
    return cls, ops


def k2_groups(ops):
    lk = ops['v_cvt_flr_i32_f32']                      # look-ups (one v_cvt_flr each)
    comb = ops['v_cvt_f32_i32_e32'] // 2               # gate registers combined (two conversions each)
    return [
        ('table look-ups: {} x 7'.format(lk), lk * 7, 'v_med3 (clamp) + v_fract + v_cvt_flr + v_lshl_add (row address) + 3 fma (cubic)'),
        ('level combine: {} gate registers x 6'.format(comb), comb * 6, '2 v_lshl_add (V, U) + 2 v_cvt_f32_i32 + fma(V, 65536, U) + fma(t, S, start)'),
        ('scalar-input start: 12 x 1', 12, 'fma(x, W, b) of layer 1'),
        ('cell update: 6 cells x 6', 36, 'fma(64, g, -32), f*C, i*G, +, fma(2, s, -1), o*th'),
        ('quantise + digits: 6 cells', ops['v_perm_b32'] + ops['v_xor_b32_e32'] + 12, 'fma + integer add per cell (q + 0x808080), v_perm_b32 per digit plane pair, xor'),
    ]


def lat_groups(ops):
    lk = ops['v_cvt_flr_i32_f32']
    return [
        ('table look-ups: {} x 7'.format(lk), lk * 7, 'as above, ONE merged gate tile per wave (three tiles in three slots)'),
        ('levels, DPP: 12 accumulator registers', ops['v_lshlrev_b32_e32'] + ops['v_add_u32_dpp'] + ops['v_cvt_f32_i32_e32'] + ops['v_mov_b32_dpp'] + 8,
         'per gate row: 3 x (v_lshl + v_add_dpp) + pack + 2 cvt + 2 (mov_dpp + fmac) + gather = 14'),
        ('scale + start: 4 x 2', 8, 'fma(t, S, fma(x, W, b))'),
        ('cell update + state select', 8, ''),
        ('quantise + digit bytes', 6, 'fma, add, xor, 2 shifts (+ 3 ds_write_b8)'),
    ]


def clamp_bounds():
    """where can the host PROVE that a look-up never leaves the table, so that its v_med3 clamp is a
    no-op?  Vector-input layers only (|h| <= 1 for every input): |u| <= g (sum |W_col| + sum |U_col| + |b|)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from poreplex_amd.config import default_config
    from poreplex_amd import native as N
    nc = N.NativeConfig(default_config(), 0)        # (owns the weight arrays the struct points at)
    st = nc.struct
    print('=' * 100)
    print('(a) clamp-free look-ups: provable bound of |u| (table units; the clamp is at [-512, 511.99997]) per gate block i, f, g, o')
    for name, kernel in (('scaler_lstm1', 'K2 layer 1'), ('scaler_lstm2', 'K2 layer 2'), ('demux_fwd', 'K5a forward'),
                         ('demux_bwd', 'K5a backward'), ('demux_top', 'K5b')):
        L = getattr(st, name)
        H, I = L.units, L.input_dim
        G = 4 * H
        k = np.ctypeslib.as_array(L.kernel, shape=(I * G,)).reshape(I, G)
        r = np.ctypeslib.as_array(L.recurrent, shape=(H * G,)).reshape(H, G)
        b = np.ctypeslib.as_array(L.bias, shape=(G,))
        scale = np.repeat([16.0, 16.0, 32.0, 16.0], H)
        if I == 1:
            print('  {:14s} {:13s} scalar input x is unbounded: no bound (recurrent part alone: {})'.format(
                name, kernel, [round(float(((np.abs(r).sum(0) + np.abs(b)) * scale)[g * H:(g + 1) * H].max()), 1) for g in range(4)]))
            continue
        tu = (np.abs(k).sum(0) + np.abs(r).sum(0) + np.abs(b)) * scale
        bounds = [round(float(tu[g * H:(g + 1) * H].max()), 1) for g in range(4)]
        print('  {:14s} {:13s} {}  -> provable for gate blocks {}'.format(name, kernel, bounds, [g for g in range(4) if bounds[g] < 511.99]))
    print('  => with the shipped weights only layer 2 of K2 has provable blocks, and only i and f (511.9 and 503.2: by a hair):'
          ' 6 of its 463 VALU per wave and step (1.3 %); K5b none.  Not built: a kernel variant per provable block for < 1 % of K2.')
    print('(b) the v_lshl_add of a look-up IS its address computation ((segment + 512) * 16 bytes): there is no separate add to fold; '
          'ds_read_b128 takes a byte address register + a 16-bit immediate, no scaled index.')
    print('(c) digits of q: already v_perm_b32 (one per digit plane and cell pair) on q + 0x808080 -- 2 VALU per cell '
          '(fma to the biased integer, one integer add) + 3 v_perm + 3 v_xor per three cells: no subtract / divide chain exists.')
    print('(d) a quadratic spline would save one fma per look-up (30 of 463) but needs ~4 096 segments for the cubic\'s 1e-8 '
          '(error ~ h^3 max|f\'\'\'| / (9 sqrt 3): 2e-6 at h = 1/16), a 64 KB table against 16 KB: K2 runs two workgroups of 81.5 KB per CU '
          'of 160 KB -- no room.  Not taken.')


if __name__ == '__main__':
    print('LSTM step-loop instruction budgets, from the ISA of this tree (tools/valu_budget.py; hipcc ' +
          subprocess.run(['/opt/rocm/bin/hipcc', '--version'], capture_output=True, text=True).stdout.splitlines()[0] + ')')
    print('cost model (profiles/r05/ubench_i8_mfma_32x32.txt, PMC passes): a SIMD issues EITHER a VALU (~3.7-4.3 cycles per wave '
          'instruction) OR an MFMA (16 cycles per v_mfma_i32_16x16x64_i8); per wave and step: cycles ~ 4 VALU + 16 MFMA')
    report('K2 k_scaler_lstm_q8 (16-read tile, 4 waves, both layers per wave; batches above 8 x #CU reads)', 'k_lstm_q8.hip', 'k_scaler_lstm_q8i', k2_groups)
    report('K2 latency form k_scaler_lstm_q8_lat (4-read tile, 8 waves: the loop holds BOTH wave kinds, layer 1 | layer 2 behind a branch)',
           'k_lstm_q8_lat.hip', 'k_scaler_lstm_q8_lat', lat_groups)
    report('K5a k_demux_bidir_q8 (16-read tile, forward + backward cell per wave)', 'k_lstm_q8_demux.hip', 'k_demux_bidir_q8i', k2_groups,
           ('-mllvm', '-amdgpu-sched-strategy=iterative-ilp'))
    report('K5b k_demux_top_q8 (16-read tile, 64-unit top cell)', 'k_lstm_q8_demux.hip', 'k_demux_top_q8i', k2_groups,
           ('-mllvm', '-amdgpu-sched-strategy=iterative-ilp'))
    clamp_bounds()
