"""Compares the dumps of two compositing-forward builds (profiles/r6_fwd_ab.py) bit for bit.
    python profiles/r6_fwd_cmp.py <tagA> <tagB>"""
import glob
import os
import sys

import torch

a, b = sys.argv[1], sys.argv[2]
bad = 0
files = sorted(glob.glob(f"/tmp/r6ab/{a}/*.pt"))
for fa in files:
    fb = fa.replace(f"/tmp/r6ab/{a}/", f"/tmp/r6ab/{b}/")
    if not os.path.exists(fb):
        print("missing", fb)
        bad += 1
        continue
    da, db = torch.load(fa), torch.load(fb)
    msg = []
    for k in ("out", "final_T", "n_contrib"):
        x, y = da[k], db[k]
        same = torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y)
        if not same:
            d = (x.double() - y.double()).abs()
            msg.append(f"{k}: {int((d > 0).sum())} of {d.numel()} differ, max |d| {float(d.max()):.3e}")
    print(os.path.basename(fa), "IDENTICAL" if not msg else "DIFFERENT " + "; ".join(msg))
    bad += bool(msg)
print(f"{a} vs {b}: {len(files) - bad} of {len(files)} dumps bit-identical")
sys.exit(1 if bad else 0)
