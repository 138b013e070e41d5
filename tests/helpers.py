"""Shared plumbing for the tests: run the same scene through the CPU oracle and the HIP library."""
from __future__ import annotations

import numpy as np
import torch

from autovfx_amd.cameras import Camera
from autovfx_amd.scenes import GaussianCloud


def oracle_kwargs(cloud: GaussianCloud, cam: Camera, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, sh_degree=None,
                  cov3D_precomp=None):
    kw = dict(means3D=cloud.means3D, opacities=cloud.opacities, bg=np.asarray(bg, np.float32),
              width=cam.image_width, height=cam.image_height, viewmatrix=cam.world_view_transform,
              projmatrix=cam.full_proj_transform, campos=cam.camera_center, tanfovx=cam.tanfovx,
              tanfovy=cam.tanfovy, sh_degree=cloud.sh_degree if sh_degree is None else sh_degree,
              scale_modifier=scale_modifier)
    if cloud.colors_precomp is not None:
        kw["colors_precomp"] = cloud.colors_precomp
    else:
        kw["shs"] = cloud.shs
    if cov3D_precomp is not None:
        kw["cov3D_precomp"] = cov3D_precomp
    else:
        kw["scales"] = cloud.scales
        kw["rotations"] = cloud.rotations
    return kw


def settings_for(cam: Camera, device, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, sh_degree=3, prefiltered=False,
                 debug=False):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=cam.tanfovx,
        tanfovy=cam.tanfovy, bg=torch.tensor(bg, dtype=torch.float32, device=device),
        scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform.to(device),
        projmatrix=cam.full_proj_transform.to(device), sh_degree=sh_degree,
        campos=cam.camera_center.to(device), prefiltered=prefiltered, debug=debug)


def run_hip(cloud: GaussianCloud, cam: Camera, device="cuda:0", bg=(0.0, 0.0, 0.0), scale_modifier=1.0,
            sh_degree=None, cov3D_precomp=None, debug=False):
    """Call the product exactly the way gaussian_renderer.render() does (reference
    gaussian_renderer/__init__.py:101-159) and return numpy outputs."""
    from diff_gaussian_rasterization import GaussianRasterizer
    c = cloud.to(device)
    st = settings_for(cam, device, bg, scale_modifier, cloud.sh_degree if sh_degree is None else sh_degree,
                      debug=debug)
    rast = GaussianRasterizer(st)
    means2D = torch.zeros_like(c.means3D)
    kw = {}
    if cov3D_precomp is not None:
        kw["cov3D_precomp"] = torch.as_tensor(cov3D_precomp, dtype=torch.float32, device=device)
    else:
        kw["scales"], kw["rotations"] = c.scales, c.rotations
    with torch.no_grad():
        color, depth, alpha, radii = rast(means3D=c.means3D, means2D=means2D, opacities=c.opacities, shs=c.shs,
                                          colors_precomp=c.colors_precomp, **kw)
    torch.cuda.synchronize()
    return {"color": color.cpu().numpy(), "depth": depth.cpu().numpy(), "alpha": alpha.cpu().numpy(),
            "radii": radii.cpu().numpy()}


def hip_forward_raw(cloud: GaussianCloud, cam: Camera, device="cuda:0", bg=(0.0, 0.0, 0.0), scale_modifier=1.0,
                    sh_degree=None, cov3D_precomp=None, debug=True, cull=False):
    """Call ``_C.rasterize_gaussians`` directly (a full, differentiable call) and decode every scratch sub-array.
    ``cull=False`` switches exact-image tile culling off so the lists are the reference's."""
    from autovfx_amd import _lib
    from diff_gaussian_rasterization import _C
    _lib.set_option(_lib.OPT_TILE_CULL, 1 if cull else 0)
    try:
        return _hip_forward_raw(_C, cloud, cam, device, bg, scale_modifier, sh_degree, cov3D_precomp, debug)
    finally:
        _lib.set_option(_lib.OPT_TILE_CULL, 1)


def hip_forward_inference(cloud: GaussianCloud, cam: Camera, device="cuda:0", bg=(0.0, 0.0, 0.0), scale_modifier=1.0,
                          sh_degree=None, cov3D_precomp=None, debug=False, slabs=0, slab_first=400, defer_colour=1,
                          slab_min_rest=0):
    """An inference call (``GSR_FORWARD_INFERENCE``: depth slabs with occlusion culling between them, colours only for
    listed splats): the public outputs and how many pairs each slab put into its list.  ``slab_min_rest=0``: the slab
    machinery runs whenever the scene is larger than two first slabs, whatever the library's pay-off threshold says."""
    from autovfx_amd import _lib
    from diff_gaussian_rasterization import _C
    _C.set_geometry_cache(False)
    _lib.set_option(_lib.OPT_SLABS, slabs)
    _lib.set_option(_lib.OPT_SLAB_FIRST, slab_first)
    _lib.set_option(_lib.OPT_DEFER_COLOUR, defer_colour)
    _lib.set_option(_lib.OPT_SLAB_MIN_REST, slab_min_rest)
    try:
        c = cloud.to(device)
        st = settings_for(cam, device, bg, scale_modifier, cloud.sh_degree if sh_degree is None else sh_degree)
        e = torch.Tensor([])
        cov = e if cov3D_precomp is None else torch.as_tensor(cov3D_precomp, dtype=torch.float32, device=device)
        (n, color, depth, alpha, radii, _g, _b, _i) = _C.rasterize_gaussians(
            st.bg, c.means3D, e if c.colors_precomp is None else c.colors_precomp, c.opacities,
            e if cov3D_precomp is not None else c.scales, e if cov3D_precomp is not None else c.rotations,
            scale_modifier, cov, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, st.image_height,
            st.image_width, e if c.shs is None else c.shs, st.sh_degree, st.campos, False, debug, inference=True)
        torch.cuda.synchronize()
        lay = _C.last_layout() if cloud.P else {"slab_pairs": []}
        return {"num_rendered": n, "color": color.cpu().numpy(), "depth": depth.cpu().numpy(), "alpha": alpha.cpu().numpy(),
                "radii": radii.cpu().numpy(), "slab_pairs": lay["slab_pairs"]}
    finally:
        _lib.set_option(_lib.OPT_SLABS, 2)
        _lib.set_option(_lib.OPT_SLAB_FIRST, 400)
        _lib.set_option(_lib.OPT_DEFER_COLOUR, 1)
        _lib.set_option(_lib.OPT_SLAB_MIN_REST, 3000000)
        _C.set_geometry_cache(None)


def _hip_forward_raw(_C, cloud, cam, device, bg, scale_modifier, sh_degree, cov3D_precomp, debug):
    c = cloud.to(device)
    st = settings_for(cam, device, bg, scale_modifier, cloud.sh_degree if sh_degree is None else sh_degree)
    e = torch.Tensor([])
    cov = e if cov3D_precomp is None else torch.as_tensor(cov3D_precomp, dtype=torch.float32, device=device)
    (n, color, depth, alpha, radii, geom, binning, img) = _C.rasterize_gaussians(
        st.bg, c.means3D, e if c.colors_precomp is None else c.colors_precomp, c.opacities,
        e if cov3D_precomp is not None else c.scales, e if cov3D_precomp is not None else c.rotations,
        scale_modifier, cov, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, st.image_height,
        st.image_width, e if c.shs is None else c.shs, st.sh_degree, st.campos, False, debug)
    torch.cuda.synchronize()
    lay = _C.last_layout() if cloud.P else {}
    out = {"num_rendered": n, "color": color.cpu().numpy(), "depth": depth.cpu().numpy(),
           "alpha": alpha.cpu().numpy(), "radii": radii.cpu().numpy()}
    if cloud.P:
        assert lay["counts"]["num_rendered"] == n
        out.update(decode_scratch(lay, geom, binning, img, cloud.P, cam.image_width, cam.image_height))
    return out


def decode_scratch(lay, geom, binning, img, P, W, H):
    """Every sub-array of the three scratch arenas of a FULL forward call (``lay`` = ``_C.last_layout()`` taken right
    after it on the calling thread), as numpy arrays."""
    T = ((W + 15) // 16) * ((H + 15) // 16)

    def view(buf, off, dtype, count, shape=None):
        nbytes = count * torch.tensor([], dtype=dtype).element_size()
        t = buf[off:off + nbytes].view(dtype)
        return (t.reshape(shape) if shape else t).cpu().numpy()

    out = {}
    g = lay["geom"]
    raster = view(geom, g["raster"], torch.float32, 8 * P, (P, 8))   # x y cxx cxy cyy opacity z pad
    out["depths"] = np.ascontiguousarray(raster[:, 6])
    out["means2D"] = np.ascontiguousarray(raster[:, 0:2])
    out["conic_opacity"] = np.ascontiguousarray(raster[:, 2:6])
    out["rgb"] = view(geom, g["rgb"], torch.float32, 3 * P, (P, 3))
    bins = view(geom, g["splat_bins"], torch.int32, 4 * P, (P, 4)).astype(np.uint32)
    out["tight_rect"] = np.stack((bins[:, 0] & 0xFFFF, bins[:, 0] >> 16, bins[:, 1] & 0xFFFF, bins[:, 1] >> 16), 1)   # x0 y0 w h
    out["live_mask"] = bins[:, 2].astype(np.uint64) | (bins[:, 3].astype(np.uint64) << np.uint64(32))
    out["depth_order"] = view(geom, g["depth_order"], torch.int32, P).astype(np.uint32)
    # The depth sort drops the Gaussians that emit nothing in its first pass (GSR_OPT_DEPTH_DROP): the order is defined for
    # the others (a record with a size) and whatever the buffer held behind them is blanked here, so that whole-array
    # comparisons between two runs compare what the library defines.
    out["sorted_count"] = int(np.count_nonzero(bins[:, 1]))
    out["depth_order"][out["sorted_count"]:] = 0xFFFFFFFF
    out["point_offsets"] = view(geom, g["point_offsets"], torch.int32, P).astype(np.uint32)
    # live tiles = pairs each splat emits: the steps of the inclusive offsets over the depth order
    tt = np.zeros(P, np.uint32)
    steps = np.diff(out["point_offsets"].astype(np.int64), prepend=0).astype(np.uint32)
    emits = steps != 0   # (positions behind the visible ones carry no pairs; the depth order there is undefined: GSR_OPT_DEPTH_DROP)
    tt[out["depth_order"][emits]] = steps[emits]
    out["tiles_touched"] = tt
    b = lay["binning"]
    live = lay["counts"]["live_pairs"]
    out["live_pairs"] = live
    out["point_list"] = view(binning, b["point_list"], torch.int32, live).astype(np.uint32)
    out["tile_keys"] = view(binning, b["tile_keys"], torch.int32, live).astype(np.uint32)
    i = lay["image"]
    out["ranges"] = view(img, i["ranges"], torch.int32, 2 * T, (T, 2)).astype(np.uint32)
    out["n_contrib"] = view(img, i["n_contrib"], torch.int32, W * H, (H, W)).astype(np.uint32)
    return out


class dump_on_failure:
    """``with dump_on_failure(name, inputs=..., hip=..., ref=...)``: when the block raises, every numpy array of the given
    dicts is saved to gpurun_out/failures/<name>.npz (merged back from the GPU box) and the failure re-raised, so a
    red run can be localised afterwards even if it never repeats."""

    def __init__(self, name, **groups):
        self.name, self.groups = name, groups

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            return False
        import os
        import traceback
        root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "failures")
        os.makedirs(root, exist_ok=True)
        flat = {}
        for gname, d in self.groups.items():
            for k, v in (d or {}).items():
                if torch.is_tensor(v):
                    v = v.detach().cpu().numpy()
                if isinstance(v, (np.ndarray, int, float, np.integer, np.floating)):
                    flat[f"{gname}.{k}"] = np.asarray(v)
        safe = "".join(ch if ch.isalnum() or ch in "-_." else "_" for ch in self.name)
        try:
            np.savez_compressed(os.path.join(root, safe + ".npz"), **flat)
            with open(os.path.join(root, safe + ".txt"), "w") as f:
                f.write("".join(traceback.format_exception(exc_type, exc, tb)))
        except OSError:
            pass
        return False


# ---- gradient parity against the fp64 truth (oracle/gsr_oracle.c: gsro_backward_f64) ----
GRAD_REL, GRAD_ABS, GRAD_REF_FACTOR = 2e-4, 1e-6, 4.0
GRAD_PER_ELEMENT = True      # the bar also holds element by element (assert_gradients_vs_truth)
GRAD_ELEMENT_FACTOR = 6.0    # ... with 6 instead of 4 times the reference's own error AT THAT ELEMENT: an element's yardstick is one
                             # sample (eight draws of the noise model, one fp32 run of the reference), the array's maximum many


def report_row(name, **kv):
    """One line of gpurun_out/parity_report.jsonl (copied to profiles/ at the end of a round)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "parity_report.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **{k: (v.item() if hasattr(v, "item") else v) for k, v in kv.items()}}) + "\n")


def gradient_errors(got, ref32, truth):
    """max-norm distances of one gradient array: (|got - truth|, |ref32 - truth|, |got - ref32|, scale = max|truth|), over the
    elements where the truth and the reference are finite (fp32 overflow in a wild scene is compared separately)."""
    g, r, t = (np.asarray(a, np.float64).reshape(-1) for a in (got, ref32, truth))
    ok = np.isfinite(t) & np.isfinite(r)
    if not ok.any():
        return 0.0, 0.0, 0.0, 0.0
    g, r, t = g[ok], r[ok], t[ok]
    return float(np.abs(g - t).max()), float(np.abs(r - t).max()), float(np.abs(g - r).max()), float(np.abs(t).max())


def assert_gradients_vs_truth(name, got, ref32, truth, keys, rel=GRAD_REL, abs_tol=GRAD_ABS, ref_factor=GRAD_REF_FACTOR, noise=None):
    """THE gradient bar.  Per gradient array, in the max norm:

        |got - truth|  <=  max(rel * max|truth| + abs_tol,  ref_factor * |ref32 - truth|)

    and, for the scenes that are ill-conditioned BY DESIGN (flat SuGaR-style Gaussians, the wild scenes), where one fp32 sample
    of the reference says little about the next, ``noise`` = cpu_oracle.fp32_noise(truth, ...) joins ``|ref32 - truth|`` under
    the factor: the reference's own fp32 chain evaluated on per-Gaussian sums that carry 2 ulp of their summands' magnitude,
    eight draws, a pure function of the scene.

    ``truth`` = the reference's backward formulas evaluated in double on the fp32 forward state (cpu_oracle.backward_f64);
    ``ref32`` = the reference's own fp32 backward (the CPU oracle, bit-identical to backward.cu compiled for the host, or the
    reference's kernels on this GPU).  The first term is the bar a well-conditioned gradient has always been held to; the
    second says an ill-conditioned one (a needle's 1 / (denom^2 + 1e-7)) may be as far from the truth as the reference
    itself is, times ``ref_factor`` -- two fp32 samples of such a gradient differ from each other by more than 2e-4 of the
    scale even with exact per-Gaussian sums (profiles/r05_gradient_truth.md).  No stragglers, no noise multipliers.
    Writes one report row with, per array: err (|got - truth| / scale), ref_err, vs_ref and frac = err / bar.

    Round 6: the max-norm bar lets ONE ill-conditioned element set the allowance of a whole array.  The same
    inequality is therefore also held ELEMENT BY ELEMENT -- element i may be as far from the truth as the fixed bar, or as
    ``ref_factor`` times the reference's own error (or the noise yardstick) AT THAT ELEMENT: a well-conditioned Gaussian inside an
    ill-conditioned array is pinned at ``rel * scale`` again.  ``frac_elem`` in the report row is the worst element's fraction."""
    row, failures = {}, []
    for k in keys:
        if got.get(k) is None:
            continue
        e_got, e_ref, e_pair, scale = gradient_errors(got[k], ref32[k], truth[k])
        e_noise = float(np.max(noise[k])) if noise is not None and k in noise and np.size(noise[k]) else 0.0
        bar = max(rel * scale + abs_tol, ref_factor * max(e_ref, e_noise))
        s = max(scale, 1e-30)
        row[k] = {"err": e_got / s, "ref_err": e_ref / s, "vs_ref": e_pair / s, "frac": e_got / bar}
        if noise is not None:
            row[k]["noise"] = e_noise / s
        if not e_got <= bar:
            failures.append(f"{k}: |got - truth| {e_got:.3e} > bar {bar:.3e} (scale {scale:.3e}, |ref32 - truth| {e_ref:.3e})")
        # ... and per element
        g_, r_, t_ = (np.asarray(a, np.float64).reshape(-1) for a in (got[k], ref32[k], truth[k]))
        ok = np.isfinite(t_) & np.isfinite(r_)
        if ok.any():
            allow = np.abs(r_ - t_)
            if noise is not None and k in noise and np.size(noise[k]) == g_.size:
                allow = np.maximum(allow, np.asarray(noise[k], np.float64).reshape(-1))
            bar_i = np.maximum(rel * scale + abs_tol, max(ref_factor, GRAD_ELEMENT_FACTOR) * allow)[ok]
            frac_i = np.abs(g_ - t_)[ok] / bar_i
            worst = int(np.argmax(frac_i))
            row[k]["frac_elem"] = float(frac_i[worst])
            if GRAD_PER_ELEMENT and not frac_i[worst] <= 1.0:
                failures.append(f"{k}: element {worst}: |got - truth| {np.abs(g_ - t_)[ok][worst]:.3e} > its own bar {bar_i[worst]:.3e} "
                                f"(scale {scale:.3e}, |ref32 - truth| there {np.abs(r_ - t_)[ok][worst]:.3e})")
    report_row("grad:" + name, **{f"{k}.{m}": v for k, d in row.items() for m, v in d.items()})
    assert not failures, f"{name}: " + "; ".join(failures)
    return row
