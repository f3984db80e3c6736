"""Does a training step leave cyclic garbage behind (objects only the interpreter's cycle collector can free)?  Tensors caught in
such cycles return to the caching allocator late and in bursts.  Prints the types found after 10 steps with the collector off."""
import gc
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import synthetic  # noqa: E402
from vcr_gaus_amd.trainer import BenchTrainer  # noqa: E402

dev = torch.device("cuda:0")
n, views, W, H, focal, sem, smult = synthetic.workload(sys.argv[1] if len(sys.argv) > 1 else "c2_dtu_300k_800x600")
bt = BenchTrainer(synthetic.make_gaussians(n, seed=0), synthetic.make_cameras(8, W, H, focal, device=dev), dev)
for i in range(10):
    bt.step(i)
torch.cuda.synchronize()
gc.collect()
gc.disable()
gc.set_debug(gc.DEBUG_SAVEALL)
for i in range(10):
    bt.step(i)
torch.cuda.synchronize()
found = gc.collect()
c = Counter(type(o).__name__ for o in gc.garbage)
tens = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
print("unreachable objects after 10 steps:", found, dict(c.most_common(12)))
print("tensors among them:", len(tens), "bytes", sum(t.numel() * t.element_size() for t in tens))
for o in gc.garbage[:0]:
    print(type(o))
