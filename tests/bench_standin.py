"""bench.py with the GPU context replaced by the oracle-backed test double (tests/oracle_context.py): what the CPU
rendezvous tests of the multi-rank driver launch (tests/test_distributed_cpu.py).  The seam lives HERE, in tests/:
bench.py has no flag or environment variable that makes it build anything but the HIP context.  Lines produced
through this wrapper say data = TEST-STANDIN and value = null."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import bench  # noqa: E402
from oracle_context import OracleBackedContext  # noqa: E402

bench.CONTEXT_CLASS = OracleBackedContext
bench.ENTRY_SCRIPT = os.path.abspath(__file__)          # a bare `--gpus N` call re-launches THIS file under torchrun

if __name__ == '__main__':
    bench.main()
