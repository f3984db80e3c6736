"""Small host-side math of `tools/general_utils.py` that is not per-Gaussian hot-path work."""
import numpy as np
import torch


def inverse_sigmoid(x):
    """`tools/general_utils.py:22-23`."""
    return torch.log(x / (1 - x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear learning-rate decay with optional warm-up (`tools/general_utils.py:49-82`)."""
    def lr_at(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        warm = 1.0
        if lr_delay_steps > 0:
            warm = lr_delay_mult + (1 - lr_delay_mult) * np.sin(
                0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        t = np.clip(step / max_steps, 0, 1)
        return warm * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
    return lr_at


def build_rotation(r):
    """Quaternion (w,x,y,z, re-normalised) -> rotation matrices [N,3,3]
    (`tools/general_utils.py:98-119`).  Used by densification (every 100 its), not per step."""
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)


def set_random_seed(seed):
    """`tools/general_utils.py:151-162` (without the hard-coded device switch)."""
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
