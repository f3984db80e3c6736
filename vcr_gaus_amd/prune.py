"""Importance / visibility accumulation over a camera stack and the volume-weighted pruning score: the operator surface
of the reference's `tools/prune.py` (`calculate_v_imp_score :6-22`, `prune_list :25-47`, `get_visi_list :50-69`) over the
forward-only raster modes of the HIP rasterizer (`count_render` f_count = 1, `visi_acc_render` f_count = 3)."""
import torch

from .gaussian_renderer import count_render, visi_acc_render


def calculate_v_imp_score(gaussians, imp_list, v_pow):
    """score_i = (volume_i / volume at the 90th percentile, counted from the largest) ** v_pow * importance_i, with
    volume = product of the three activated scales (`tools/prune.py:6-22`)."""
    volume = torch.prod(gaussians.get_scaling, dim=1)
    kth = torch.sort(volume, descending=True)[0][int(len(volume) * 0.9)]
    return torch.pow(volume / kth, v_pow) * imp_list


@torch.no_grad()
def prune_list(gaussians, viewpoint_stack, pipe, background):
    """Per-Gaussian hit count and sum of alpha * T over all cameras of `viewpoint_stack` (consumed like the reference
    does: the list is emptied)."""
    count = score = None
    while viewpoint_stack:
        pkg = count_render(viewpoint_stack.pop(), gaussians, pipe, background)
        count = pkg["gaussians_count"] if count is None else count + pkg["gaussians_count"]
        score = pkg["important_score"] if score is None else score + pkg["important_score"]
    return count, score


@torch.no_grad()
def get_visi_list(gaussians, viewpoint_stack, pipe, background):
    """{"visi": [N] bool}: Gaussians that contributed to at least one pixel of at least one camera."""
    count = None
    while viewpoint_stack:
        c = visi_acc_render(viewpoint_stack.pop(), gaussians, pipe, background)["countlist"]
        count = c if count is None else count + c
    return {"visi": count > 0}
