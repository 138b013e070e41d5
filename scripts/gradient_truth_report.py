"""profiles/r05_gradient_truth.md: who is how far from the gradient?

For BASELINE-sized and ill-conditioned scenes: this library's backward (float atomics, and GSR_OPT_BACKWARD_DETERMINISTIC), the CPU
oracle's fp32 backward (= backward.cu compiled for the host, bit for bit) and the reference's own kernels on this GPU
(oracle/_ref/libgsr_ref_hip.so), each against the fp64 truth (oracle/gsr_oracle.c: gsro_backward_f64).  Per gradient array the
max-norm distance over the scale of the array; per Gaussian how often this library is further from the truth than the
reference.  Run on the GPU box:  python scripts/gradient_truth_report.py > gpurun_out/gradient_truth.md
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from autovfx_amd import _lib, scenes                         # noqa: E402
from autovfx_amd.cameras import Camera, orbit_cameras, sugar_orbit_cameras   # noqa: E402
from autovfx_amd.scenes import GaussianCloud                 # noqa: E402
from oracle import cpu_oracle, ref_hip                       # noqa: E402
from helpers import gradient_errors, oracle_kwargs           # noqa: E402
from test_oracle_backward import GOLDEN_BW, load_bw_case, pixel_grads   # noqa: E402

DEV = torch.device("cuda", 0)
KEYS = ("dL_dmeans2D", "dL_dmeans3D", "dL_dopacity", "dL_dcolors", "dL_dsh", "dL_dscales", "dL_drotations")
REFKEY = {"dL_dmeans3D": "means3D", "dL_dopacity": "opacity", "dL_dsh": "sh", "dL_dscales": "scales", "dL_drotations": "rotations",
          "dL_dmeans2D": "means2D", "dL_dcolors": "colors"}


def hip_grads(cloud, cam, bg, pg, mode, tanfov=None):
    from diff_gaussian_rasterization import GaussianRasterizer
    from helpers import settings_for
    c = cloud.to(DEV)
    st = settings_for(cam, DEV, tuple(float(v) for v in bg), 1.0, cloud.sh_degree)
    if tanfov is not None:
        st = st._replace(tanfovx=tanfov[0], tanfovy=tanfov[1])
    leaf = lambda t: None if t is None else t.clone().requires_grad_(True)
    L = dict(means3D=leaf(c.means3D), opac=leaf(c.opacities), shs=leaf(c.shs), colors=leaf(c.colors_precomp), scales=leaf(c.scales),
             rots=leaf(c.rotations), m2d=torch.zeros_like(c.means3D, requires_grad=True))
    _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 1 if mode == "deterministic" else 0)
    try:
        color, depth, alpha, radii = GaussianRasterizer(st)(means3D=L["means3D"], means2D=L["m2d"], opacities=L["opac"], shs=L["shs"],
                                                            colors_precomp=L["colors"], scales=L["scales"], rotations=L["rots"])
        t = lambda a: torch.from_numpy(np.asarray(a)).to(DEV)
        ((color * t(pg["dL_dcolor"])).sum() + (depth * t(pg["dL_ddepth"])).sum() + (alpha * t(pg["dL_dalpha"])).sum()).backward()
        torch.cuda.synchronize()
    finally:
        _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 0)
    g = lambda x: None if x is None or x.grad is None else x.grad.cpu().numpy()
    return {"dL_dmeans3D": g(L["means3D"]), "dL_dmeans2D": g(L["m2d"]), "dL_dopacity": g(L["opac"]), "dL_dsh": g(L["shs"]),
            "dL_dcolors": g(L["colors"]), "dL_dscales": g(L["scales"]), "dL_drotations": g(L["rots"])}


def ref_gpu_grads(cloud, cam, bg, pg):
    if not ref_hip.available():
        return None
    c, cm = cloud.to(DEV), cam.to(DEV)
    b = torch.tensor([float(v) for v in bg], device=DEV)
    n, _c, _d, a_ref, r_ref = ref_hip.forward(c, cm, b)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(DEV)
    g = ref_hip.backward(c, cm, b, n, r_ref, a_ref, t(pg["dL_dcolor"]), t(pg["dL_ddepth"]), t(pg["dL_dalpha"]))
    return {k: g[v].cpu().numpy() for k, v in REFKEY.items()}


def report(name, cloud, cam, bg, pg, tanfov=None):
    kw = oracle_kwargs(cloud, cam, bg=np.asarray(bg, np.float32))
    if tanfov is not None:
        kw["tanfovx"], kw["tanfovy"] = tanfov
    kw.update(pg)
    t0 = time.time()
    o32, truth = cpu_oracle.backward(**kw), cpu_oracle.backward_f64(**kw)
    t_cpu = time.time() - t0
    samples = {"hip atomic": hip_grads(cloud, cam, bg, pg, "atomic", tanfov), "hip atomic, 2nd run": hip_grads(cloud, cam, bg, pg, "atomic", tanfov),
               "hip deterministic": hip_grads(cloud, cam, bg, pg, "deterministic", tanfov), "oracle fp32 (CPU)": o32}
    if tanfov is None:
        rg = ref_gpu_grads(cloud, cam, bg, pg)
        if rg is not None:
            samples["reference kernels (GPU)"] = rg
    P = cloud.P
    print(f"\n### {name}  (P = {P}, {cam.image_width}x{cam.image_height}, oracles {t_cpu:.1f} s)\n")
    print("max |x - truth| / max |truth| per gradient array:\n")
    keys = [k for k in KEYS if samples["hip atomic"].get(k) is not None]
    print("| sample | " + " | ".join(k[3:] for k in keys) + " |")
    print("|---|" + "---|" * len(keys))
    for sname, s in samples.items():
        cells = []
        for k in keys:
            if s.get(k) is None:
                cells.append("-")
                continue
            e, _, _, sc = gradient_errors(s[k].reshape(truth[k].shape), o32[k], truth[k])
            cells.append(f"{e / max(sc, 1e-30):.2e}")
        print(f"| {sname} | " + " | ".join(cells) + " |")
    # bar and fraction
    print("\nbar = max(2e-4 scale + 1e-6, 4 |oracle fp32 - truth|); `hip atomic` as a fraction of it: " + ", ".join(
        f"{k[3:]} {gradient_errors(samples['hip atomic'][k].reshape(truth[k].shape), o32[k], truth[k])[0] / max(2e-4 * gradient_errors(o32[k], o32[k], truth[k])[3] + 1e-6, 4 * gradient_errors(o32[k], o32[k], truth[k])[0]):.2f}"
        for k in keys))
    # per Gaussian: hip vs the reference sample, distance to truth
    print("\nper Gaussian (touched ones: truth != 0), distance to the truth in the max norm over the Gaussian's row:\n")
    print("| array | touched | hip further than oracle fp32 | ... by more than 2x | median hip / oracle | worst hip / scale | worst oracle / scale | hip(atomic) vs hip(det) / scale | two atomic runs / scale |")
    print("|---|---|---|---|---|---|---|---|---|")
    for k in keys:
        t = truth[k].reshape(P, -1)
        fin = np.isfinite(t).all(1)
        d = lambda a: np.abs(np.nan_to_num(np.asarray(a, np.float64).reshape(P, -1) - t, nan=0.0, posinf=0.0, neginf=0.0)).max(1)
        dh, do = d(samples["hip atomic"][k]), d(o32[k])
        touched = fin & (np.abs(np.nan_to_num(t)).max(1) > 0)
        n = int(touched.sum())
        if n == 0:
            continue
        sc = float(np.abs(t[fin]).max())
        ratio = dh[touched] / np.maximum(do[touched], 1e-300)
        pair = np.abs(np.nan_to_num(samples["hip atomic"][k].astype(np.float64) - samples["hip deterministic"][k])).max() / sc
        rerun = np.abs(np.nan_to_num(samples["hip atomic"][k].astype(np.float64) - samples["hip atomic, 2nd run"][k])).max() / sc
        print(f"| {k[3:]} | {n} | {100.0 * float((dh[touched] > do[touched]).mean()):.1f} % | {100.0 * float((dh[touched] > 2 * do[touched]).mean()):.1f} % | "
              f"{float(np.median(ratio)):.2f} | {dh[touched].max() / sc:.2e} | {do[touched].max() / sc:.2e} | {pair:.2e} | {rerun:.2e} |")
    sys.stdout.flush()


def main():
    print("# Gradient truth report\n")
    print(__doc__.split("Run on the GPU box")[0].strip())
    print(f"\ndevice: {torch.cuda.get_device_name(0)}; host threads {os.cpu_count()}")
    cloud, cam = scenes.config_c4(), sugar_orbit_cameras(50, 960, 540)[25]
    report("c4_full_200k (BASELINE configs[3]: flat SuGaR-style Gaussians, thin axis 3e-4)", cloud, cam, (1.0, 1.0, 1.0), pixel_grads(cam, 8))
    cloud, cam = scenes.config_c2(), orbit_cameras(200, 960, 540)[100]
    report("c2_full_1M (BASELINE configs[1] stand-in)", cloud, cam, (0.0, 0.0, 0.0), pixel_grads(cam, 5))
    if "--quick" not in sys.argv:
        cloud, cam = scenes.config_c3(), orbit_cameras(800, 1920, 1080)[400]
        report("c3_full_3M (BASELINE configs[2])", cloud, cam, (0.0, 0.0, 0.0), pixel_grads(cam, 5))
    for path in GOLDEN_BW:
        kw, _ref = load_bw_case(path)
        t = lambda k: None if k not in kw else torch.from_numpy(np.asarray(kw[k]))
        cloud = GaussianCloud(t("means3D"), t("opacities"), t("scales"), t("rotations"), t("shs"), t("colors_precomp"), kw["sh_degree"])
        cam = Camera(kw["width"], kw["height"], 2 * np.arctan(kw["tanfovx"]), 2 * np.arctan(kw["tanfovy"]), t("viewmatrix"), t("projmatrix"),
                     t("projmatrix"), t("campos"))
        if float(kw["scale_modifier"]) != 1.0:
            continue
        report("golden " + os.path.basename(path)[:-4], cloud, cam, tuple(float(v) for v in kw["bg"]),
               {k: kw[k] for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")}, tanfov=(kw["tanfovx"], kw["tanfovy"]))
    from test_reference_hip_gpu import wild_training_case
    for seed in range(6):
        cloud, cam, bg, (w_c, w_d, w_a) = wild_training_case(seed, DEV)
        pg = {"dL_dcolor": w_c.cpu().numpy(), "dL_ddepth": w_d.cpu().numpy(), "dL_dalpha": w_a.cpu().numpy()}
        report(f"wild seed {seed} (tests/test_reference_hip_gpu.py: needles 1e-4 .. 10, near-plane splats, unnormalised quaternions)",
               cloud.to("cpu"), cam.to("cpu"), tuple(float(v) for v in bg.cpu()), pg)


if __name__ == "__main__":
    main()
