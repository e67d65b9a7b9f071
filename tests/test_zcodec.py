"""Compressed samples (include/pxg.h, pxg_z_*): the host codec round-trips every int16 pattern,
a compressed read bundle is the same bundle, and the GPU decoder (-m gpu) fills the same arena
the host decoder does."""
import os

import numpy as np
import pytest

from poreplex_amd import native as N
from poreplex_amd.fast5_file import ReadBundle
from poreplex_amd.synth import synth_batch
from poreplex_amd.worker_persistence import WorkerPersistenceStorage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(params=['packed', 'bytes'])
def codec(request):
    """Both encodings of the deltas (include/pxg.h: PXG_Z_PACKED, the default of ABI 5, and PXG_Z_BYTES, what
    older bundles hold)."""
    return request.param


def adversarial_reads(rng):
    parts = []
    for n in (0, 1, 2, 7, 1023, 1024, 1025, 2047, 2048, 2049, 5000, 0, 3):
        kind = len(parts) % 5
        if kind == 0:
            x = rng.integers(-32768, 32768, n)                         # every delta needs two bytes
        elif kind == 1:
            x = np.full(n, -32768)                                      # zero deltas, extreme first
        elif kind == 2:
            x = np.where(np.arange(n) % 2 == 0, 32767, -32768)          # deltas that wrap around
        elif kind == 3:
            x = 500 + np.cumsum(rng.integers(-127, 128, n))             # one byte each, both signs: zz <= 255 edge
        else:
            x = 500 + np.cumsum(rng.choice([-128, 127, 128, -129, 0], n))   # straddles the 1 / 2 byte boundary
        parts.append(np.asarray(x).astype(np.int16))
    return parts


def test_codec_round_trip_and_chunk_layout(codec):
    rng = np.random.default_rng(4)
    parts = adversarial_reads(rng)
    arena, off = N.pack_reads(parts)
    z, chunks, base = N.z_encode(arena, off, codec)
    assert (chunks['codec'] == N.Z_CODECS[codec]).all()
    assert np.array_equal(N.z_decode(z, chunks, len(arena)), arena)
    # chunks never span reads; chunk_base indexes them by read
    per_read = (np.diff(off) + N.Z_CHUNK - 1) // N.Z_CHUNK
    assert np.array_equal(np.diff(base), per_read) and len(chunks) == per_read.sum()
    for r in range(len(parts)):
        c = chunks[base[r]:base[r + 1]]
        assert c['len'].sum() == len(parts[r])
        if len(c):
            assert c['dst'][0] == off[r] and np.array_equal(c['first'], arena[c['dst']])
    # one byte per sample where the deltas are small, two where they are not (packed: groups of four round up)
    assert len(z) <= len(chunks) * (N.Z_CHUNK // 8 + 8) + 2 * len(arena)
    sb = synth_batch(8, seed=3, samples_per_read=30000)
    z2, ch2, _ = N.z_encode(sb['arena'], sb['offsets'], codec)
    assert np.array_equal(N.z_decode(z2, ch2, len(sb['arena'])), sb['arena'])
    # bytes per sample on bench-like signal: ~1.17 as byte codes, ~0.96 bit-packed
    assert len(z2) < (1.3 if codec == 'bytes' else 1.0) * len(sb['arena'])


def test_one_stream_may_mix_the_codecs_and_garbage_widths_stay_inside_the_stream():
    """The codec is a property of the CHUNK: a stream whose first reads were written as byte codes and the rest
    bit-packed decodes as one.  A packed chunk's widths come from the stream, so the host decoder is told where
    the stream ends and reads nothing behind it, whatever the header says."""
    sb = synth_batch(6, seed=12, samples_per_read=5000)
    o = sb['offsets']
    za, ca, _ = N.z_encode(sb['arena'][:o[3]], o[:4], 'bytes')
    zb, cb, _ = N.z_encode(sb['arena'][o[3]:], o[3:] - o[3], 'packed')
    cb = cb.copy()
    cb['data_off'] += len(za)
    cb['dst'] += o[3]
    z, chunks = np.concatenate([za, zb]), np.concatenate([ca, cb])
    assert np.array_equal(N.z_decode(z, chunks, len(sb['arena'])), sb['arena'])
    bad = z.copy()
    last = int(chunks['data_off'][-1])
    bad[last:last + 128] = 0xFF                      # every group of the last chunk claims 16 bits
    out = N.z_decode(bad, chunks, len(sb['arena']))  # wrong samples in that chunk, no stray read
    assert np.array_equal(out[:chunks['dst'][-1]], sb['arena'][:chunks['dst'][-1]])


def compressed_copy(src, dst, codec='packed'):
    """The same bundle with its arena replaced by the encoded form."""
    with np.load(src, allow_pickle=False) as npz:
        d = {k: npz[k] for k in npz.files}
    z, chunks, base = N.z_encode(d.pop('arena'), d['offsets'], codec)
    d.update(arena_z=z, z_chunks=chunks, z_chunk_base=base, bundle_version=np.int64(3))
    np.savez(dst, **d)
    return dst


@pytest.mark.parametrize('bundle', ['batch0.pxr.npz', 'chimera.pxr.npz'])
def test_compressed_bundle_is_the_same_bundle(tmp_path, bundle, codec):
    plain = ReadBundle(os.path.join(GOLDEN, bundle))
    comp = ReadBundle(compressed_copy(os.path.join(GOLDEN, bundle), str(tmp_path / bundle), codec))
    assert comp.compressed and not plain.compressed
    n = len(plain.read_ids)
    for i in range(n):
        assert np.array_equal(comp.samples(i), plain.samples(i))
    run = comp.samples_run(2, min(7, n))
    assert isinstance(run, N.EncodedSamples)
    assert np.array_equal(run.decode(), plain.samples_run(2, min(7, n)))
    assert len(comp.samples_run(3, 3)) == 0


@pytest.mark.parametrize('bundle', ['batch0.pxr.npz', 'chimera.pxr.npz'])
def test_session_from_a_compressed_bundle_writes_the_same_files(tmp_path, monkeypatch, bundle):
    """Session driver over the compressed bundle (oracle test double decodes with the host
    reference decoder where the GPU context decodes on the device): byte-identical outputs."""
    from oracle_context import OracleBackedContext
    from test_session import session_config, golden_reads
    from poreplex_amd.session import GpuSession
    monkeypatch.setattr(N, 'NativeContext', OracleBackedContext)
    outs = {}
    for tag, path in (('plain', os.path.join(GOLDEN, bundle)),
                      ('z', compressed_copy(os.path.join(GOLDEN, bundle), str(tmp_path / bundle)))):
        WorkerPersistenceStorage.reset()
        cfg = session_config(tmp_path / tag, bundle)
        cfg['read_bundle'] = path
        outs[tag] = GpuSession(cfg, batch_reads=6).run(golden_reads(bundle))
        WorkerPersistenceStorage.reset()
    assert (tmp_path / 'plain' / 'sequencing_summary.txt').read_bytes() == \
        (tmp_path / 'z' / 'sequencing_summary.txt').read_bytes()
    assert outs['plain']['labels'].tobytes() == outs['z']['labels'].tobytes()
    assert np.array_equal(outs['plain']['counts'], outs['z']['counts'])


@pytest.mark.gpu
def test_device_decoder_fills_the_arena_the_host_decoder_does(ctx, oracle, codec):
    """pxg_batch_stage_z: bytes + chunk records cross the link, k_z_decode rebuilds the int16
    arena; every stage then gives the records of the uncompressed upload -- for adversarial
    sample patterns and for a slice of a larger encoded stream (non-zero bases)."""
    rng = np.random.default_rng(5)
    sb = synth_batch(40, seed=11, samples_per_read=30000, jitter=0.4)
    o = sb['offsets']
    parts = [sb['arena'][o[i]:o[i + 1]] for i in range(40)] + adversarial_reads(rng)
    arena, off = N.pack_reads(parts)
    cal = np.concatenate([sb['calib'], np.repeat(sb['calib'][:1], len(parts) - 40)])
    z, chunks, base = N.z_encode(arena, off, codec)
    mask = N.STAGE_ALL_DEMUX | N.STAGE_POLYA
    ctx.upload(arena, off, cal)
    ctx.run(mask)
    want = ctx.download().copy()
    ctx.stage_z(N.EncodedSamples(z, chunks, 0, 0, len(arena)), off, cal)
    ctx.swap()
    ctx.run(mask)
    got = ctx.download()
    for f in got.dtype.names:
        assert np.array_equal(got[f], want[f], equal_nan=True), f
    assert np.array_equal(got['status'], oracle.process_batch(arena, off, cal, None, mask)['status'])
    # reads 5 .. 25 as a slice of the stream
    i0, i1 = 5, 25
    c0, c1 = base[i0], base[i1]
    enc = N.EncodedSamples(z[chunks['data_off'][c0]:chunks['data_off'][c1]], chunks[c0:c1],
                           chunks['data_off'][c0], off[i0], off[i1] - off[i0])
    ctx.stage_z(enc, off[i0:i1 + 1] - off[i0], cal[i0:i1])
    ctx.swap()
    ctx.run(mask)
    got = ctx.download()
    for f in got.dtype.names:
        assert np.array_equal(got[f], want[f][i0:i1], equal_nan=True), f


@pytest.mark.gpu
def test_fuzz_device_decoder_sample_for_sample(ctx, codec):
    """Random ragged batches of every sample pattern: the arena k_z_decode leaves in HBM equals
    the original, sample for sample (also through a slice with non-zero bases)."""
    rng = np.random.default_rng(int(os.environ.get('PXG_FUZZ_SEED', 31)))
    for trial in range(int(os.environ.get('PXG_FUZZ_TRIALS', 12))):
        parts = []
        for _ in range(int(rng.integers(1, 40))):
            n = int(rng.choice([0, 1, 2, 15, 16, 17, 1023, 1024, 1025, 4095, 4096, 4097])) if rng.random() < 0.5 \
                else int(rng.integers(0, 70000))
            kind = rng.integers(0, 5)
            if kind == 0:
                x = rng.integers(-32768, 32768, n)
            elif kind == 1:
                x = 500 + np.cumsum(rng.integers(-130, 131, n))
            elif kind == 2:
                x = np.full(n, int(rng.integers(-32768, 32768)))
            elif kind == 3:
                x = np.where(rng.random(n) < 0.01, 32767, -32768)
            else:
                x = 600 + 80 * np.sin(np.arange(n) / 7.0) + rng.normal(0, 12, n)
            parts.append(np.asarray(x).astype(np.int16))
        if sum(len(p) for p in parts) == 0:
            parts.append(np.arange(5, dtype=np.int16))
        arena, off = N.pack_reads(parts)
        cal = np.zeros(len(parts), dtype=N.CALIB_DTYPE)
        cal['range'], cal['digitisation'], cal['sampling_rate'] = 1400.0, 8192.0, 3012.0
        z, chunks, base = N.z_encode(arena, off, codec)
        ctx.stage_z(N.EncodedSamples(z, chunks, 0, 0, len(arena)), off, cal)
        ctx.swap()
        assert np.array_equal(ctx.download_samples(len(arena)), arena), trial
        i0 = int(rng.integers(0, len(parts)))
        i1 = int(rng.integers(i0 + 1, len(parts) + 1))
        if off[i1] - off[i0] > 0:
            c0, c1 = int(base[i0]), int(base[i1])
            b0 = int(chunks['data_off'][c0])
            b1 = int(chunks['data_off'][c1]) if c1 < len(chunks) else len(z)
            enc = N.EncodedSamples(z[b0:b1], chunks[c0:c1], b0, int(off[i0]), int(off[i1] - off[i0]))
            ctx.stage_z(enc, off[i0:i1 + 1] - off[i0], cal[i0:i1])
            ctx.swap()
            assert np.array_equal(ctx.download_samples(len(enc)), arena[off[i0]:off[i1]]), (trial, i0, i1)


def _encoded(seed=11, codec='packed'):
    sb = synth_batch(6, seed=seed, samples_per_read=5000)
    z, chunks, base = N.z_encode(sb['arena'], sb['offsets'], codec)
    return sb, z, chunks, base


CORRUPTIONS = {
    'length zero': lambda z, c: c['len'].__setitem__(2, 0),
    'length beyond a chunk': lambda z, c: c['len'].__setitem__(2, 2000),
    'destination moved': lambda z, c: c['dst'].__setitem__(3, c['dst'][3] + 7),
    'destination before the arena': lambda z, c: c['dst'].__setitem__(0, -5),
    'bytes before the stream': lambda z, c: c['data_off'].__setitem__(0, -128),
    'bytes beyond the stream': lambda z, c: c['data_off'].__setitem__(len(c) - 1, len(z) + 4096),
    'chunks overlap in the stream': lambda z, c: c['data_off'].__setitem__(4, c['data_off'][3]),
    'unknown codec': lambda z, c: c['codec'].__setitem__(1, 7),
}


@pytest.mark.parametrize('what', sorted(CORRUPTIONS))
def test_corrupt_chunk_records_are_refused_on_the_host(what, codec):
    """chunk records come from files -- a truncated or corrupt bundle must be an error
    (PXG_E_INVALID / PxgError), never an out-of-bounds access in a decoder."""
    sb, z, chunks, base = _encoded(codec=codec)
    N.z_validate(z, chunks, len(sb['arena']))                  # the encoder's own output passes
    bad = chunks.copy()
    CORRUPTIONS[what](z, bad)
    with pytest.raises(N.PxgError):
        N.z_decode(z, bad, len(sb['arena']))
    with pytest.raises(N.PxgError):                            # a truncated byte stream
        N.z_decode(z[:len(z) // 2], chunks, len(sb['arena']))
    with pytest.raises(N.PxgError):                            # records that do not cover the arena
        N.z_decode(z, chunks[:-1], len(sb['arena']))


def test_truncated_compressed_bundle_fails_at_load(tmp_path):
    from poreplex_amd.fast5_file import write_bundle
    sb, z, chunks, base = _encoded()
    n = len(sb['offsets']) - 1
    path = str(tmp_path / 'ok.pxr.npz')
    write_bundle(path, sb['arena'], sb['offsets'], sb['calib'], ['f%d' % i for i in range(n)],
                 ['r%d' % i for i in range(n)], compress=True)
    ReadBundle(path)
    with np.load(path, allow_pickle=False) as npz:
        d = {k: npz[k] for k in npz.files}
    for name, change in (('cut', lambda d: d.__setitem__('arena_z', d['arena_z'][:1000])),
                         ('records', lambda d: d.__setitem__('z_chunks', d['z_chunks'][:-3])),
                         ('span', lambda d: d['z_chunks']['dst'].__setitem__(slice(1, None), d['z_chunks']['dst'][1:] + 1))):
        e = {k: v.copy() for k, v in d.items()}
        change(e)
        bad = str(tmp_path / (name + '.pxr.npz'))
        np.savez(bad, **e)
        with pytest.raises(N.PxgError):
            ReadBundle(bad)


@pytest.mark.gpu
@pytest.mark.parametrize('what', sorted(CORRUPTIONS))
def test_corrupt_chunk_records_are_refused_by_stage_z(ctx, what):
    sb, z, chunks, base = _encoded()
    bad = chunks.copy()
    CORRUPTIONS[what](z, bad)
    with pytest.raises(N.PxgError, match='chunk records'):
        ctx.stage_z(N.EncodedSamples(z, bad, 0, 0, len(sb['arena'])), sb['offsets'], sb['calib'])
    with pytest.raises(N.PxgError, match='chunk records'):
        ctx.stage_z(N.EncodedSamples(z[:len(z) // 2], chunks, 0, 0, len(sb['arena'])), sb['offsets'], sb['calib'])
    # and the context is still usable: the good stream decodes to the arena
    ctx.stage_z(N.EncodedSamples(z, chunks, 0, 0, len(sb['arena'])), sb['offsets'], sb['calib'])
    ctx.swap()
    assert np.array_equal(ctx.download_samples(len(sb['arena'])), sb['arena'])


@pytest.mark.gpu
def test_garbage_width_headers_stay_inside_the_stream_on_the_device(ctx):
    """A packed chunk's widths are data: with every header nibble of the LAST chunk of the stream claiming 16 bits
    the device decoder reads no byte behind the stream's buffer (clamped staging) -- that chunk's samples are
    wrong, every other chunk's are right, and the context carries on."""
    sb, z, chunks, base = _encoded(codec='packed')
    bad = z.copy()
    last = int(chunks['data_off'][-1])
    bad[last:last + 128] = 0xFF
    ctx.stage_z(N.EncodedSamples(bad, chunks, 0, 0, len(sb['arena'])), sb['offsets'], sb['calib'])
    ctx.swap()
    got = ctx.download_samples(len(sb['arena']))
    cut = int(chunks['dst'][-1])
    assert np.array_equal(got[:cut], sb['arena'][:cut])
    ctx.stage_z(N.EncodedSamples(z, chunks, 0, 0, len(sb['arena'])), sb['offsets'], sb['calib'])
    ctx.swap()
    assert np.array_equal(ctx.download_samples(len(sb['arena'])), sb['arena'])
