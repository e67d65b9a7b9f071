"""Helper of test_gpu_parity.py::test_label_all_gather_over_rccl_single_rank (not a test module)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (torch + RCCL first, like bench.py for N > 1)
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))

from poreplex_amd import native as N  # noqa: E402
from poreplex_amd.config import default_config  # noqa: E402
from poreplex_amd.distributed import count_table, gather_labels, label_records, reduce_counts  # noqa: E402

ctx = N.NativeContext(default_config(), device_id=0)
b = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'batch0.pxr.npz')))
res = ctx.process_batch(b['arena'], b['offsets'], b['calib'])
want = label_records(res)
assert (res['status'] == 0).sum() > 10
for sizes in (None, [len(res)]):
    got = gather_labels(res, dist, sizes=sizes, force=True)
    assert got.tobytes() == want.tobytes()
t = torch.from_numpy(count_table(want)).cuda()
dist.all_reduce(t)
assert np.array_equal(t.cpu().numpy(), count_table(want))
dist.barrier()
torch.cuda.synchronize()
ctx.close()
dist.destroy_process_group()
print('RCCL-SINGLE-RANK-OK')
