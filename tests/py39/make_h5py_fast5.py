"""Runs under an interpreter that has h5py (the image's /opt/conda/bin/python3.9): writes FAST5
files with the REAL HDF5 library in the layouts the native reader (csrc/pxg_h5.cpp) must read
-- single / multi-read, contiguous / chunked + gzip (+ shuffle, fletcher32), fixed and
variable-length string attributes, an Events table, a 700-read file (multi-level group B-tree),
and one `libver latest` file the reader must DECLINE -- plus the truth as .npy files.
usage: make_h5py_fast5.py <outdir>"""
import h5py, numpy as np, sys, os
rng = np.random.default_rng(0)
out = sys.argv[1]
def fill(node_raw, node_ch, node_tr, analyses, rid, raw, i, bc=True, vlen=False, events=False, **dskw):
    enc = (lambda s: s) if vlen else (lambda s: s.encode())
    node_raw.attrs['duration'] = np.uint32(len(raw)); node_raw.attrs['start_time'] = np.uint64(1000*i+7)
    node_raw.attrs['read_id'] = enc(rid); node_raw.attrs['read_number'] = np.int32(i)
    node_raw.create_dataset('Signal', data=raw, **dskw)
    node_ch.attrs['channel_number'] = enc(str(100+i)); node_ch.attrs['digitisation'] = 8192.0
    node_ch.attrs['offset'] = float(3+i); node_ch.attrs['range'] = 1200.5; node_ch.attrs['sampling_rate'] = 3012.0
    node_tr.attrs['run_id'] = enc('run' + 'ab'*18); node_tr.attrs['sample_id'] = enc('sampleX')
    for k in range(12): node_tr.attrs['extra%d' % k] = enc('v%d' % k)
    if not bc: return
    g = analyses.create_group('Basecall_1D_000'); t = g.create_group('BaseCalled_template')
    seq = ''.join(rng.choice(list('ACGU'), 50+i)); q = ''.join(chr(40+int(x)) for x in rng.integers(0,20,50+i))
    t.create_dataset('Fastq', data=np.string_('@%s\n%s\n+\n%s\n' % (rid, seq, q)) if not vlen else '@%s\n%s\n+\n%s\n' % (rid, seq, q))
    mv = (rng.random(len(raw)//15) < 0.1).astype(np.uint8)
    if events:
        ev = np.zeros(len(mv), dtype=[('model_state','S5'),('move','u1'),('p_model_state','<f4')]); ev['move']=mv; ev['p_model_state']=rng.random(len(mv))
        t.create_dataset('Events', data=ev)
    else:
        t.create_dataset('Move', data=mv, compression='gzip', chunks=(97,))
    s = g.create_group('Summary/basecall_1d_template'); s.attrs['sequence_length']=np.int32(50+i); s.attrs['mean_qscore']=np.float32(9.123+i); s.attrs['block_stride']=np.int32(15)
    sg = analyses.create_group('Segmentation_000/Summary/segmentation'); sg.attrs['num_events_template']=np.int32(len(mv)); sg.attrs['first_sample_template']=np.int32(i)
    np.save(os.path.join(out, 'truth_%s.npy' % rid), {'raw': raw, 'seq': seq, 'q': q, 'mv': mv, 'pms': ev['p_model_state'] if events else None}, allow_pickle=True)
def raw_of(i): return (500 + np.cumsum(rng.integers(-30, 31, 5000 + 137*i))).astype(np.int16)
with h5py.File(os.path.join(out,'single.fast5'),'w') as h5:
    fill(h5.create_group('Raw/Reads/Read_7'), h5.create_group('UniqueGlobalKey/channel_id'), h5.create_group('UniqueGlobalKey/tracking_id'), h5.create_group('Analyses'), 'rid-single', raw_of(0), 0)
with h5py.File(os.path.join(out,'single_events.fast5'),'w') as h5:
    fill(h5.create_group('Raw/Reads/Read_8'), h5.create_group('UniqueGlobalKey/channel_id'), h5.create_group('UniqueGlobalKey/tracking_id'), h5.create_group('Analyses'), 'rid-events', raw_of(1), 1, events=True, compression='gzip', shuffle=True, chunks=(1500,))
# albacore's 14-column Events table (uint64 start / length as albacore writes them), chunked + gzip
ALB = [('mean','<f4'),('start','<u8'),('stdv','<f4'),('length','<u8'),('model_state','S5'),('move','<i4'),('weights','<f4'),
       ('p_model_state','<f4'),('mp_state','S5'),('p_mp_state','<f4'),('p_A','<f4'),('p_C','<f4'),('p_G','<f4'),('p_T','<f4')]
with h5py.File(os.path.join(out,'single_albacore.fast5'),'w') as h5:
    raw = raw_of(3)
    fill(h5.create_group('Raw/Reads/Read_10'), h5.create_group('UniqueGlobalKey/channel_id'), h5.create_group('UniqueGlobalKey/tracking_id'), h5.create_group('Analyses'), 'rid-albacore', raw, 3, bc=False)
    g = h5['Analyses'].create_group('Basecall_1D_000'); t = g.create_group('BaseCalled_template')
    n = 611
    ev = np.zeros(n, dtype=ALB)
    ev['length'] = rng.integers(2, 15, n); ev['start'] = 5 + np.concatenate([[0], np.cumsum(ev['length'][:-1])])
    ev['mean'] = rng.normal(90, 12, n); ev['stdv'] = rng.random(n) * 3; ev['move'] = rng.integers(0, 3, n)
    ev['model_state'] = np.array([''.join(k) for k in rng.choice(list('ACGT'), (n, 5))], dtype='S5'); ev['mp_state'] = ev['model_state']
    ev['p_model_state'] = rng.random(n); ev['weights'] = 1
    seq = ''.join(rng.choice(list('ACGU'), int(ev['move'].sum()))); q = ''.join(chr(40+int(x)) for x in rng.integers(0,20,len(seq)))
    t.create_dataset('Fastq', data=np.string_('@rid-albacore\n%s\n+\n%s\n' % (seq, q)))
    t.create_dataset('Events', data=ev, chunks=(100,), compression='gzip')
    s_ = g.create_group('Summary/basecall_1d_template'); s_.attrs['sequence_length']=np.int32(len(seq)); s_.attrs['mean_qscore']=np.float32(9.5)
    sg = h5['Analyses'].create_group('Segmentation_000/Summary/segmentation'); sg.attrs['num_events_template']=np.int32(n); sg.attrs['first_sample_template']=np.int32(5)
    np.save(os.path.join(out, 'truth_albacore_events.npy'), ev)
# the same table with an 11-character model_state column (wider than the 5-mers of the shipped models)
ALBW = [(k, 'S11' if k in ('model_state', 'mp_state') else d) for k, d in ALB]
with h5py.File(os.path.join(out,'single_albacore_wide.fast5'),'w') as h5:
    fill(h5.create_group('Raw/Reads/Read_11'), h5.create_group('UniqueGlobalKey/channel_id'), h5.create_group('UniqueGlobalKey/tracking_id'), h5.create_group('Analyses'), 'rid-albacore-wide', raw, 3, bc=False)
    g = h5['Analyses'].create_group('Basecall_1D_000'); t = g.create_group('BaseCalled_template')
    evw = np.zeros(n, dtype=ALBW)
    for k, _d in ALB:
        evw[k] = ev[k]
    evw['model_state'] = np.array([''.join(k) for k in rng.choice(list('ACGT'), (n, 11))], dtype='S11'); evw['mp_state'] = evw['model_state']
    t.create_dataset('Fastq', data=np.string_('@rid-albacore-wide\n%s\n+\n%s\n' % (seq, q)))
    t.create_dataset('Events', data=evw)
    s_ = g.create_group('Summary/basecall_1d_template'); s_.attrs['sequence_length']=np.int32(len(seq)); s_.attrs['mean_qscore']=np.float32(9.5)
    sg = h5['Analyses'].create_group('Segmentation_000/Summary/segmentation'); sg.attrs['num_events_template']=np.int32(n); sg.attrs['first_sample_template']=np.int32(5)
    np.save(os.path.join(out, 'truth_albacore_wide_events.npy'), evw)
with h5py.File(os.path.join(out,'single_latest.fast5'),'w', libver='latest') as h5:
    fill(h5.create_group('Raw/Reads/Read_9'), h5.create_group('UniqueGlobalKey/channel_id'), h5.create_group('UniqueGlobalKey/tracking_id'), h5.create_group('Analyses'), 'rid-latest', raw_of(2), 2, bc=False)
with h5py.File(os.path.join(out,'multi.fast5'),'w') as h5:
    for i in range(10, 22):
        b = h5.create_group('read_rid-m%02d' % i)
        fill(b.create_group('Raw'), b.create_group('channel_id'), b.create_group('tracking_id'), b.create_group('Analyses'), 'rid-m%02d' % i, raw_of(i), i, vlen=(i%2==0), bc=(i!=13), compression='gzip', chunks=(4096,), fletcher32=(i%3==0))
with h5py.File(os.path.join(out,'many.fast5'),'w') as h5:
    for i in range(700):
        b = h5.create_group('read_many%04d' % i)
        fill(b.create_group('Raw'), b.create_group('channel_id'), b.create_group('tracking_id'), b.create_group('Analyses'), 'many%04d' % i, raw_of(i%7)[:600], i, bc=False)
print('ok')
