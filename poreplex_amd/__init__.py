"""poreplex_amd -- MI355X-native raw-signal hot path for poreplex.

Keeps the reference's per-read processor surface (``process_batch``,
``SignalAnalyzer``, ``SignalAnalysis``; poreplex/signal_analyzer.py:38) and
runs every numeric stage as hand-written HIP kernels behind the C ABI in
``include/pxg.h`` (``poreplex_amd/csrc/libpxg.so``).
"""
__version__ = '0.1.0'

# output-name constants the result records are routed by
# (reference: poreplex/__init__.py:30-38, commandline.py:137-159)
OUTPUT_NAME_PASSED = 'pass'
OUTPUT_NAME_FAILED = 'fail'
OUTPUT_NAME_ARTIFACT = 'artifact'
OUTPUT_NAME_BARCODES = 'BC{n}'
OUTPUT_NAME_UNDETERMINED = 'undetermined'
