"""Plain attribute-dict configs with the reference's key names (`configs/config_base.yaml`,
`configs/reconstruct.yaml`, `configs/{dtu,tnt,360_v2}/base.yaml`; effective values resolved in
SURVEY.md Appendix C).  The reference's YAML/CLI machinery itself is out of scope."""
import copy


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(d):
    return AttrDict({k: _wrap(v) if isinstance(v, dict) else v for k, v in d.items()})


_BASE = {
    "seed": 0,
    "model": {"sh_degree": 3, "white_background": False, "use_decoupled_appearance": False, "enable_semantic": False,
              "ch_sem_feat": 0, "num_cls": 0, "max_mem": 22, "depth_type": "intersection", "sphere": False},
    "optim": {
        "iterations": 30000, "position_lr_init": 0.00016, "position_lr_final": 0.0000016,
        "position_lr_delay_mult": 0.01, "position_lr_max_steps": 30000, "feature_lr": 0.0025, "opacity_lr": 0.05,
        "scaling_lr": 0.005, "rotation_lr": 0.001, "cls_lr": 0.0005, "percent_dense": 0.01, "densification_interval": 100,
        "opacity_reset_interval": 3000, "densify_from_iter": 500, "densify_until_iter": 15000,
        "densify_grad_threshold": 0.0005, "random_background": False, "mask_depth_thr": 0.0, "exp_t": 0.01,
        "normal_from_iter": 0, "dnormal_from_iter": 0, "consistent_normal_from_iter": 0, "close_depth_from_iter": 0,
        "loss_weight": {"l1": 0.8, "ssim": 0.2, "l1_scale": 1.0, "mono_normal": 0.01, "depth_normal": 0.0,
                        "consistent_normal": 0.0, "distortion": 0.0, "depth_var": 0.0, "semantic": 0.0,
                        "mono_depth": 0.0, "entropy": 0.0},
        "densify_large": {"percent_dense": 0.0, "sample_cams": {"random": True, "num": 0, "up": False, "around": True}},
        "prune": {"iterations": [], "percent": 0.5, "decay": 0.6, "v_pow": 0.1},
    },
    "pipline": {"convert_SHs_python": False, "compute_cov3D_python": False, "debug": False},
}

# The reference's EFFECTIVE configurations (configs/config.py resolving configs/{dtu/dtu_scan24,tnt/Barn,360_v2/base}.yaml over
# reconstruct.yaml and config_base.yaml), pinned by tests/golden/g9_effective_configs.json (test_oracle_cpu.py).
_PRESETS = {
    "dtu": {"optim": {"exp_t": 0.01, "mask_depth_thr": 0.0, "random_background": False,
                      "consistent_normal_from_iter": 15000, "close_depth_from_iter": 15000,
                      "loss_weight": {"depth_normal": 0.0, "consistent_normal": 0.05, "mono_normal": 0.01, "distortion": 1000.0},
                      "densify_large": {"percent_dense": 1e-2, "sample_cams": {"random": False, "num": 30}},
                      "prune": {"iterations": [15000, 25000]}}},
    "tnt": {"optim": {"exp_t": 0.005, "mask_depth_thr": 0.8, "random_background": True,
                      "loss_weight": {"depth_normal": 0.015, "consistent_normal": 0.0, "mono_normal": 0.01, "semantic": 0.005},
                      "densify_large": {"percent_dense": 2e-3, "sample_cams": {"random": True, "num": 200}},
                      "prune": {"iterations": [15000, 25000]}}},
    "360": {"optim": {"exp_t": 0.01, "mask_depth_thr": 1.0, "random_background": True,
                      "loss_weight": {"depth_normal": 0.01, "consistent_normal": 0.0, "mono_normal": 0.01},
                      "densify_large": {"percent_dense": 5e-2, "sample_cams": {"random": False, "num": 100}},
                      "prune": {"iterations": [15000, 25000]}}},
    # BASELINE.json config 3 ("DTU-shape scene, full training loop with D-Normal + normal-consistency losses"): the dtu preset
    # with both normal losses on from the first iteration and the late-phase distortion loss off
    "dtu_c3": {"optim": {"exp_t": 0.01, "mask_depth_thr": 0.0, "random_background": False,
                         "loss_weight": {"depth_normal": 0.015, "consistent_normal": 0.05, "mono_normal": 0.01},
                         "densify_large": {"percent_dense": 1e-2, "sample_cams": {"random": False, "num": 30}},
                         "prune": {"iterations": [15000, 25000]}}},
}


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def make_config(preset="tnt", **overrides):
    d = copy.deepcopy(_BASE)
    _merge(d, _PRESETS[preset])
    _merge(d, overrides)
    return _wrap(d)
