"""Size-independent properties at BASELINE.json's full metric size (1 M Gaussians, 1920x1080), where the
oracle is too slow to run: linearity in colour, alpha = 1 - T_final via the background term, determinism of
the forward, and the identity  sum_i dL/dcolour_i = sum_pixels alpha  that ties backward to forward."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(device):
    from oracle import model_torch as OM
    from vcr_gaus_amd import synthetic
    raw = synthetic.make_gaussians(1_000_000, seed=0)
    cam = synthetic.make_cameras(8, 1920, 1080, 1165.0, device=device)[3]
    act = OM.activations(raw)
    return cam, {k: v.to(device).contiguous() for k, v in act.items()}


def run(cam, act, colors, bg, device, f_count=0, requires_grad=False, normals=None):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    s = GaussianRasterizationSettings(image_height=1080, image_width=1920, tanfovx=math.tan(cam.FoVx * 0.5),
                                      tanfovy=math.tan(cam.FoVy * 0.5), bg=bg.to(device), scale_modifier=1.0,
                                      viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3,
                                      campos=cam.camera_center, prefiltered=False, debug=False, f_count=f_count)
    c = colors.clone().requires_grad_(requires_grad)
    res = GaussianRasterizer(s)(means3D=act["xyz"], means2D=torch.zeros_like(act["xyz"]), colors_precomp=c,
                                opacities=act["opacity"], scales=act["scaling"], rotations=act["rotation"],
                                normals_precomp=normals)
    return res, c


def test_full_size_properties(device, scene):
    cam, act = scene
    N = act["xyz"].shape[0]
    g = torch.Generator().manual_seed(0)
    c1 = torch.rand(N, 3, generator=g).to(device)
    c2 = torch.rand(N, 3, generator=g).to(device)
    zero = torch.zeros(3)
    (o1, radii), _ = run(cam, act, c1, zero, device)
    (o2, _), _ = run(cam, act, c2, zero, device)
    (o12, _), cg = run(cam, act, c1 + c2, zero, device, requires_grad=True)
    # determinism of the forward (no atomics): bit-identical on repeat
    (o1b, _), _ = run(cam, act, c1, zero, device)
    assert torch.equal(o1, o1b)
    # linearity in colour; all non-colour channels unaffected
    assert torch.allclose(o12[:3], o1[:3] + o2[:3], rtol=1e-4, atol=1e-5)
    assert torch.equal(o1[3:], o2[3:])
    # background enters as T_final * bg  ->  alpha = 1 - T_final
    (ow, _), _ = run(cam, act, c1, torch.ones(3), device)
    T = ow[0] - o1[0]
    assert torch.allclose(1.0 - T, o1[7], atol=2e-5)
    assert float(o1[7].min()) >= 0.0 and float(o1[7].max()) <= 1.0 + 1e-5
    # constant camera-space normal n for every Gaussian -> normal channels = n * alpha
    nrm = torch.tensor([[0.0, 0.6, 0.8]], device=device).expand(N, 3).contiguous()
    (on, _), _ = run(cam, act, c1, zero, device, normals=nrm)
    assert torch.allclose(on[5], 0.6 * on[7], atol=1e-5) and torch.allclose(on[6], 0.8 * on[7], atol=1e-5)
    # backward: d(sum of red)/d colour_i,red = sum_pixels w_i ; summed over i = sum_pixels alpha
    o12[0].sum().backward()
    tot = float(cg.grad[:, 0].double().sum())
    ref = float(o12[7].double().sum())
    assert abs(tot - ref) <= 1e-4 * ref
    assert float(cg.grad[:, 1:].abs().max()) == 0.0
    assert int((radii > 0).sum()) > 0
    # count mode: sum_i score_i = sum_pixels alpha; counted Gaussians are visible ones
    (cnt, score, img, r2), _ = run(cam, act, c1, zero, device, f_count=1)
    assert abs(float(score.double().sum()) - ref) <= 1e-4 * ref
    assert torch.equal(r2, radii) and bool(((cnt > 0) <= (radii > 0)).all())
    assert torch.allclose(img, o1[:3], atol=1e-6)
