import sys, os, torch
os.environ["VCR_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcr_gaus_amd import synthetic, rasterizer
from vcr_gaus_amd.config import make_config
from vcr_gaus_amd.gaussian_model import GaussianModel
from vcr_gaus_amd.gaussian_renderer import render
from vcr_gaus_amd.graphics_utils import get_all_px_dir
dev = torch.device('cuda:0')
raw = synthetic.make_gaussians(1_000_000, seed=0)
cams = synthetic.make_cameras(8, 1920, 1080, 1165.0, device=dev)
cfg = make_config('tnt')
m = GaussianModel(cfg.model); m.create_from_params(raw, 1.0, device=dev); m.active_sh_degree = 3; m.extent = 3.3
dirs = get_all_px_dir(cams[0].intr, 1080, 1920)
for c in cams[:3]:
    for rep in range(2):
        with torch.no_grad():
            pkg = render(c, m, cfg, torch.zeros(3, device=dev), dirs=dirs)
    torch.cuda.synchronize()
    t4 = pkg["raster"].timing.view(torch.int64).view(-1, 4).cpu()
    t4 = t4[t4[:, 1] > 0]
    order = (t4[:, 1] - t4[:, 0]).argsort(descending=True)[:3]
    for o in order:
        r = t4[o]
        print("  heavy wave: dur_us=%.0f chunks=%d survivors=%d hit_survivors=%d" % ((r[1]-r[0])/100.0, r[2] >> 32, r[2] & 0xFFFFFFFF, r[3]))
    print("  totals: chunks=%d survivors=%d hit=%d" % ((t4[:,2] >> 32).sum(), (t4[:,2] & 0xFFFFFFFF).sum(), t4[:,3].sum()))
    dur4 = (t4[:, 1] - t4[:, 0]).double() / 100.0
    surv = (t4[:, 2] & 0xFFFFFFFF).double(); chunks = (t4[:, 2] >> 32).double()
    st = (t4[:, 0] - t4[:, 0].min()).double() / 100.0
    for lo, hi in [(0, 10), (10, 50), (50, 100), (100, 200), (200, 400), (400, 10000)]:
        sel = (surv >= lo) & (surv < hi)
        if sel.any():
            print(f"  survivors[{lo},{hi}): n={int(sel.sum())} dur_mean={dur4[sel].mean():.1f}us dur_max={dur4[sel].max():.1f} us/surv={(dur4[sel].sum()/surv[sel].sum().clamp_min(1)):.2f} chunks_mean={chunks[sel].mean():.1f} start_mean={st[sel].mean():.1f}us")
    tc = (t4[:, 3] >> 32).double() / 100.0; ts = (t4[:, 3] & 0xFFFFFFFF).double() / 100.0
    busy = surv >= 10
    print(f"  busy waves: dur_mean={dur4[busy].mean():.1f}us cull+wait_mean={tc[busy].mean():.1f}us surv_mean={ts[busy].mean():.1f}us other={(dur4[busy]-tc[busy]-ts[busy]).mean():.1f}us; per-chunk cull+wait={(tc[busy].sum()/chunks[busy].sum()):.2f}us per-survivor={(ts[busy].sum()/surv[busy].sum()):.3f}us")
    late = st > 100
    print(f"  waves starting after 100us: n={int(late.sum())} dur_mean={dur4[late].mean():.1f} us/surv={(dur4[late].sum()/surv[late].sum()):.2f};  before: us/surv={(dur4[~late].sum()/surv[~late].sum()):.2f}")
    t = t4[:, :2]
    dur = (t[:, 1] - t[:, 0]).double()
    span = float(t[:, 1].max() - t[:, 0].min())
    srt = dur.sort(descending=True).values
    print(f"waves={len(dur)} span_ticks={span:.0f} max_wave={srt[0]:.0f} ({srt[0]/span:.2f} of span) top5={srt[:5].tolist()} "
          f"mean={dur.mean():.0f} p99={srt[int(0.01*len(srt))]:.0f} sum/span={dur.sum()/span:.0f} waves-in-flight avg")
