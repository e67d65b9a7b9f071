"""FAST5 input (h5py) is only importable under /opt/conda/bin/python3.9 in this image; these
tests run the checks there through a subprocess and skip cleanly where that interpreter, its
h5py, or (for the reader-vs-reference comparison) /root/reference does not exist."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY39 = '/opt/conda/bin/python3.9'


def need_py39_h5py():
    if not os.path.exists(PY39):
        pytest.skip('no ' + PY39)
    if subprocess.run([PY39, '-c', 'import h5py, numpy, yaml, scipy'], capture_output=True).returncode != 0:
        pytest.skip('python3.9 lacks h5py / numpy / yaml / scipy')


def test_fast5_reader_equals_the_reference_reader():
    """Fast5Reader / get_read_ids on single- and multi-read FAST5 == the REAL reference
    classes (metadata, raw -> pA, basecall summary)."""
    need_py39_h5py()
    if not os.path.isdir('/root/reference/poreplex'):
        pytest.skip('reference tree not present')
    out = subprocess.run([PY39, os.path.join(ROOT, 'tools', 'validate_fast5_reader.py')],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert 'differences: 0' in out.stdout


def test_session_from_fast5_files_equals_bundle():
    """enumerate_reads + the per-read FAST5 path of the loader thread + the session driver:
    same sequencing_summary.txt, labels and count table as the bundle path."""
    need_py39_h5py()
    out = subprocess.run([PY39, os.path.join(ROOT, 'tests', 'py39', 'fast5_session_check.py')],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.strip().splitlines()[-1].startswith('OK 14 reads')
