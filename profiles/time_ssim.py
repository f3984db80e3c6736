"""l1 + ssim forward / backward kernel time at 1080p for the library VCR_LIB points to (HIP events, 200 repetitions)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import _lib  # noqa: E402
from vcr_gaus_amd.loss_utils import l1_ssim  # noqa: E402

dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
g = torch.Generator(device="cpu").manual_seed(0)
a = torch.rand(3, H, W, generator=g).to(dev).requires_grad_(True)
b = torch.rand(3, H, W, generator=g).to(dev)
for _ in range(5):
    l1, ss = l1_ssim(a, b)
    (l1 + ss).backward()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(200):
    ev[0].record()
    l1, ss = l1_ssim(a, b)
    ev[1].record()
    (l1 + ss).backward()
    ev[2].record()
    torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print(f"{os.path.basename(_lib.LIB_PATH)} {W}x{H}: l1_ssim forward {1e3 * tf / 200:.1f} us, backward {1e3 * tb / 200:.1f} us (event-timed, includes the small finalize / seed kernels), "
      f"l1 {float(l1):.6f} ssim {float(ss):.6f}")
