#!/opt/conda/bin/python3.9
"""Goldens for the result sinks: run the REFERENCE's own
SequencingSummaryWriter / FinalSummaryTracker / FASTQWriter (poreplex/io.py) and
setup_output_name_mapping (poreplex/commandline.py) on the result dicts the real
process_batch produced (tests/golden/batch0.results.json, chimera.results.json)
and record their outputs in tests/golden/sinks.json.

Runs under /opt/conda/bin/python3.9 (pandas + h5py live there).  pysam is not
installed: BGZFile is stubbed with gzip (the golden keeps the DECOMPRESSED FASTQ
text, so the container format does not matter); pipeline / alignment_writer are
stubbed because commandline.py imports them and they need mappy / asyncio plumbing.
"""
import gzip
import io
import json
import os
import sys
import tempfile
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
GOLDEN = os.path.join(REPO, 'tests', 'golden')
TMP = tempfile.mkdtemp(prefix='pxg_sinks_')


def build_shadow():
    pk = os.path.join(TMP, 'poreplex')
    os.makedirs(pk)
    for f in os.listdir(REF + '/poreplex'):
        if f.endswith('.py') and f not in ('pipeline.py', 'alignment_writer.py'):
            os.symlink(os.path.join(REF, 'poreplex', f), os.path.join(pk, f))
    with open(os.path.join(pk, 'pipeline.py'), 'w') as fh:
        fh.write('class ProcessingSession:\n    pass\n')
    with open(os.path.join(pk, 'alignment_writer.py'), 'w') as fh:
        fh.write('def check_minimap2_index(*a, **k):\n    return True\n')
    pysam = types.ModuleType('pysam')
    pysam.BGZFile = lambda path, mode='r': gzip.open(path, mode + 'b' if 'b' not in mode else mode)
    pysam.faidx = lambda *a, **k: None
    sys.modules['pysam'] = pysam
    sys.path.insert(0, TMP)


def dicts_of(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        doc = json.load(fh)
    out = []
    for r in doc['results']:
        r = dict(r)
        if 'sequence' in r:
            r['sequence'] = tuple(r['sequence'])
        out.append(r)
    return doc, out


def main():
    build_shadow()
    from poreplex import io as RIO
    from poreplex.commandline import setup_output_name_mapping
    cases = {}
    for tag, fname, flags in (
            ('batch0', 'batch0.results.json', {'filter_unsplit_reads': False}),
            ('chimera', 'chimera.results.json', {'filter_unsplit_reads': True}),
            ('mixed', 'batch0.results.json', {'filter_unsplit_reads': True})):
        doc, results = dicts_of(fname)
        if tag == 'mixed':
            # the same real result dicts, three times over, with barcode calls spread over
            # BC1-BC4 / undetermined and a few artifacts, so that every column and both
            # sort keys of the final summary are exercised
            big = []
            for rep in range(3):
                for i, r in enumerate(results):
                    r = dict(r)
                    r['read_id'] = '%s-%d' % (r.get('read_id', 'x'), rep) if 'read_id' in r else None
                    if r['read_id'] is None:
                        del r['read_id']
                    if r.get('label') == 'pass':
                        k = (i * 7 + rep * 3) % 6
                        if k < 4:
                            r['barcode'], r['barcode_guess'], r['barcode_score'] = k, k, 10 + k
                        if (i + rep) % 9 == 0:
                            r['status'], r['label'] = 'unsplit_read', 'artifact'
                    big.append(r)
            results = big
        config = {'barcoding': True, 'measure_polya': tag != 'chimera', 'fast5_output': tag == 'mixed',
                  'filter_unsplit_reads': flags['filter_unsplit_reads'],
                  'demultiplexing': {'number_of_barcodes': 4}}
        label_names, barcode_names, layout = setup_output_name_mapping(config)
        outdir = os.path.join(TMP, tag)
        os.makedirs(outdir)
        w = RIO.SequencingSummaryWriter(config, outdir, label_names, barcode_names)
        w.write_results(results)
        w.close()
        with open(os.path.join(outdir, 'sequencing_summary.txt')) as fh:
            summary_txt = fh.read()
        trk = RIO.FinalSummaryTracker(label_names, barcode_names)
        trk.feed_results(results)
        buf = io.StringIO()
        trk.print_results(buf)
        fq = RIO.FASTQWriter(outdir, layout)
        fq.write_sequences(results)
        fq.close()
        fastq = {}
        for key, name in layout.items():
            path = os.path.join(outdir, 'fastq', name + '.fastq.gz')
            with gzip.open(path, 'rt') as fh:
                fastq[name] = fh.read()
        # with barcoding off / polya on too
        config2 = dict(config, barcoding=False)
        ln2, bn2, layout2 = setup_output_name_mapping(config2)
        out2 = os.path.join(TMP, tag + '_nobc')
        os.makedirs(out2)
        results2 = [{k: v for k, v in r.items() if not k.startswith('barcode')} for r in results]
        w2 = RIO.SequencingSummaryWriter(config2, out2, ln2, bn2)
        w2.write_results(results2)
        w2.close()
        with open(os.path.join(out2, 'sequencing_summary.txt')) as fh:
            summary2 = fh.read()
        trk2 = RIO.FinalSummaryTracker(ln2, bn2)
        trk2.feed_results(results2)
        buf2 = io.StringIO()
        trk2.print_results(buf2)
        cases[tag] = {
            'source': fname, 'config': config, 'results': results if tag == 'mixed' else None,
            'label_names': label_names,
            'barcode_names': [[k, v] for k, v in barcode_names.items()],
            'layout': [[list(k), v] for k, v in layout.items()],
            'sequencing_summary': summary_txt, 'final_summary': buf.getvalue(), 'fastq': fastq,
            'nobarcoding': {'label_names': ln2, 'barcode_names': [[k, v] for k, v in bn2.items()],
                            'layout': [[list(k), v] for k, v in layout2.items()],
                            'sequencing_summary': summary2, 'final_summary': buf2.getvalue()},
        }
        print(tag, len(results), 'results;', len(summary_txt.splitlines()), 'summary lines')
        print(buf.getvalue())
    with open(os.path.join(GOLDEN, 'sinks.json'), 'w') as fh:
        json.dump(cases, fh, indent=1)
    print('wrote', os.path.join(GOLDEN, 'sinks.json'))


if __name__ == '__main__':
    main()
