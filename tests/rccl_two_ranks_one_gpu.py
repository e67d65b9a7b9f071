"""Helper of test_gpu_parity.py::test_rccl_with_two_ranks_on_one_device (not a test module): rank RANK of
WORLD_SIZE 2, BOTH on GPU 0, backend nccl (= RCCL).  If the runtime accepts two ranks on one device the label
all-gather and the count all-reduce are driven through it; if it refuses (RCCL normally rejects duplicate
devices) the process prints what it said, for the test to record."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

rank = int(os.environ['RANK'])
torch.cuda.set_device(0)
try:
    dist.init_process_group('nccl', rank=rank, world_size=2, device_id=torch.device('cuda', 0))
    t = torch.ones(4, device='cuda') * (rank + 1)
    dist.all_reduce(t)                     # the first collective creates the communicator
    torch.cuda.synchronize()
except Exception as exc:                   # the runtime's own words
    print('RCCL-TWO-RANKS-REFUSED: {}: {}'.format(type(exc).__name__, str(exc).replace('\n', ' ')[:400]))
    sys.exit(0)
assert t.tolist() == [3.0] * 4

from poreplex_amd import native as N  # noqa: E402
from poreplex_amd.config import default_config  # noqa: E402
from poreplex_amd.distributed import count_table, gather_labels, label_records, shard_range  # noqa: E402

ctx = N.NativeContext(default_config(), device_id=0)
b = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'batch0.pxr.npz')))
n = len(b['offsets']) - 1
lo, hi = shard_range(n, rank, 2)
o = b['offsets']
res = ctx.process_batch(b['arena'][o[lo]:o[hi]], o[lo:hi + 1] - o[lo], b['calib'][lo:hi])
sizes = [shard_range(n, r, 2)[1] - shard_range(n, r, 2)[0] for r in range(2)]
got = gather_labels(res, dist, first_index=lo, sizes=sizes)
assert len(got) == n and sorted(got['read_index'].tolist()) == list(range(n))
c = torch.from_numpy(count_table(label_records(res, first_index=lo))).cuda()
dist.all_reduce(c)
assert int(c.sum().item()) == n
dist.barrier()
torch.cuda.synchronize()
ctx.close()
dist.destroy_process_group()
print('RCCL-TWO-RANKS-OK')
