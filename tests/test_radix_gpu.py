"""The hand-written radix sort (gsr_radix.hip) against torch's stable sort, through the C ABI."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


RANK_FORMS = {"ballots": (0, 0), "verified_lds_adds": (1, 1), "lds_adds_with_injected_inversions": (3, 2)}   # request, ACTIVE


@pytest.fixture(params=list(RANK_FORMS), autouse=True)
def rank_variant(request):
    """All in-wave rank forms of the scatter kernel are compiled in (gsr_radix.hip); every test of this file runs with each of
    them forced (GSR_OPT_RADIX_RANK 0 / 1 / 3), then the default (2: verified LDS adds where the self-test passed) is restored.
    Form 3 swaps two ranks in every wave: the kernel's order check has to catch it and the ballots repair it, so the same
    assertions hold."""
    from autovfx_amd import _lib
    req, active = RANK_FORMS[request.param]
    _lib.set_option(_lib.OPT_RADIX_RANK, req)
    assert _lib.get_option(_lib.OPT_RADIX_RANK_ACTIVE) == active
    yield request.param
    _lib.set_option(_lib.OPT_RADIX_RANK, 2)


def fallbacks():
    from autovfx_amd import _lib
    torch.cuda.synchronize()
    n = _lib.get_option(_lib.OPT_RADIX_RANK_FALLBACKS)
    assert n >= 0, "the fallback counter could not be read"
    return n


def sort_pairs(keys, vals, bits, iota=False):
    from autovfx_amd import _lib
    n = keys.numel()
    dev = keys.device
    k_alt, v_alt = torch.empty_like(keys), torch.empty_like(keys)
    v = vals if vals is not None else torch.empty_like(keys)
    nbytes = int(_lib.lib.gsr_radix_scratch_bytes(n, bits))
    scratch = torch.randint(0, 255, (max(nbytes, 1),), dtype=torch.uint8, device=dev)   # garbage on purpose
    where = ctypes.c_int(0)
    rc = _lib.lib.gsr_radix_sort_pairs(n, bits, keys.data_ptr(), k_alt.data_ptr(), v.data_ptr(), v_alt.data_ptr(),
                                       int(iota), scratch.data_ptr(), nbytes, ctypes.byref(where),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    return (k_alt, v_alt) if where.value else (keys, v)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 4095, 4096, 4097, 8192, 12345, 100_000, 1_000_003, 8_500_000])
@pytest.mark.parametrize("bits", [13, 32])
def test_sort_matches_stable_sort(n, bits):
    g = torch.Generator(device="cuda").manual_seed(n * 31 + bits)
    hi = (1 << bits) - 1
    keys = torch.randint(0, min(hi, 2**31 - 1) + 1, (n,), generator=g, device="cuda", dtype=torch.int64)
    if bits == 32:                       # cover the top bit and heavy duplicates
        keys = keys * 2 + torch.randint(0, 2, (n,), generator=g, device="cuda", dtype=torch.int64)
        keys[::7] = keys[0].clone()
    vals = torch.randint(0, 2**31 - 1, (n,), generator=g, device="cuda", dtype=torch.int64)
    order = torch.sort(keys, stable=True).indices
    want_k, want_v = keys[order], vals[order]
    k32 = keys.to(torch.uint32).view(torch.int32).contiguous()
    v32 = vals.to(torch.int32).contiguous()
    got_k, got_v = sort_pairs(k32, v32, bits)
    assert torch.equal(got_k.view(torch.uint32).to(torch.int64), want_k)
    assert torch.equal(got_v.to(torch.int64), want_v)


@pytest.mark.parametrize("n", [5, 4096 * 3 + 17, 3_000_000])
def test_iota_payload_gives_the_stable_permutation(n):
    g = torch.Generator(device="cuda").manual_seed(n)
    depth = torch.rand(n, generator=g, device="cuda") * 50 + 0.2
    depth[::5] = depth[1].clone()                                  # ties: broken by index
    keys = depth.view(torch.int32).clone()
    keys[::11] = -1                                          # kCulledKey sorts last
    want = torch.sort(keys.view(torch.uint32).to(torch.int64), stable=True).indices
    _, got = sort_pairs(keys.clone(), None, 32, iota=True)
    assert torch.equal(got.to(torch.int64), want)


def test_few_distinct_digits_and_repeated_calls():
    """Every key in one digit bin (the look-back carries whole tiles), and state reuse across calls."""
    n = 300_000
    for rep in range(3):
        keys = torch.full((n,), 7 + rep, device="cuda", dtype=torch.int32)
        vals = torch.arange(n, device="cuda", dtype=torch.int32)
        k, v = sort_pairs(keys, vals, 13)
        assert torch.equal(v, torch.arange(n, device="cuda", dtype=torch.int32)) and bool((k == 7 + rep).all())


def test_lds_atomics_serve_lanes_in_order():
    """What the scatter kernel's ranking relies on (gsr_radix.hip): lanes of one wave64 instruction that add to the same
    LDS counter with a returning atomic get the counter's values in ascending lane order.  2048 workgroups x 4 waves x
    512 instructions, from all 64 lanes on one counter to all on different ones: no lane may see anything else."""
    import ctypes
    from autovfx_amd import _lib
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for seed in (1, 2, 3):
        assert _lib.lib.gsr_selftest_lds_atomic_order(2048, 512, seed, bad.data_ptr(), stream) == 0, _lib.last_error()
    torch.cuda.synchronize()
    assert int(bad.item()) == 0


def test_selftested_mode_tests_the_device_and_reports_what_it_uses():
    """GSR_OPT_RADIX_RANK = 2 (the default): the first sort on a device runs the lane-order self-test; on MI355X it passes and
    GSR_OPT_RADIX_RANK_ACTIVE says verified LDS adds.  (Were it to fail, ACTIVE would say ballots and the sorts would still be
    right: that branch is the `ballots` parametrisation of every test above.)"""
    from autovfx_amd import _lib
    _lib.set_option(_lib.OPT_RADIX_RANK, 0)
    assert _lib.get_option(_lib.OPT_RADIX_RANK_ACTIVE) == 0
    _lib.set_option(_lib.OPT_RADIX_RANK, 2)
    assert _lib.get_option(_lib.OPT_RADIX_RANK) == 2
    assert _lib.get_option(_lib.OPT_RADIX_RANK_ACTIVE) == 1, "the LDS lane-order self-test failed on this device"
    with pytest.raises(RuntimeError):
        _lib.set_option(_lib.OPT_RADIX_RANK_ACTIVE, 0)
    with pytest.raises(RuntimeError):
        _lib.set_option(_lib.OPT_RADIX_RANK_FALLBACKS, 0)
    keys = torch.randint(0, 2**13, (1_000_003,), device="cuda", dtype=torch.int32)
    vals = torch.arange(keys.numel(), device="cuda", dtype=torch.int32)
    want = torch.sort(keys.to(torch.int64), stable=True).indices.to(torch.int32)
    _, got = sort_pairs(keys.clone(), vals, 13)
    assert torch.equal(got, want)


def test_the_order_check_catches_inversions_and_stays_silent_otherwise(rank_variant):
    """The scatter kernel's own check of the LDS adds' ranks (gsr_radix.hip step 4b): with an inversion injected into every wave
    every tile must notice and repair it (GSR_OPT_RADIX_RANK_FALLBACKS counts tiles x passes); with the plain adds on this
    hardware, and with ballots, no tile may."""
    n = 4096 * 50
    # 13 key bits = a 7-bit and a 6-bit pass; four values per digit, so that in every wave some lane shares lane 0's digit
    keys = (torch.randint(0, 4, (n,), device="cuda") + 128 * torch.randint(0, 4, (n,), device="cuda")).to(torch.int32)
    vals = torch.arange(n, device="cuda", dtype=torch.int32)
    want = torch.sort(keys.to(torch.int64), stable=True).indices.to(torch.int32)
    before = fallbacks()
    _, got = sort_pairs(keys.clone(), vals, 13)
    assert torch.equal(got, want)
    grew = fallbacks() - before
    if rank_variant == "lds_adds_with_injected_inversions":
        assert grew == 50 * 2, grew      # two passes of 50 tiles, every one of them repaired
    else:
        assert grew == 0, grew


def test_first_sort_inside_a_graph_capture_does_not_run_the_selftest(rank_variant):
    """A fresh process whose FIRST sort is recorded into a hipGraph (default rank request 2, device not tested yet): the lane-order
    self-test allocates and synchronises, neither of which is legal during capture -- the library must skip it, capture the ballot
    form, and test the device later, outside the capture.  (Run once: the parametrisation does not reach the child process.)"""
    if rank_variant != "ballots":
        pytest.skip("one run is enough: the child process uses the library default")
    import os, subprocess, sys
    code = r"""
import ctypes, torch
from autovfx_amd import _lib
assert _lib.get_option(_lib.OPT_RADIX_RANK) == 2
n, bits = 200_003, 13
keys = torch.randint(0, 2**bits, (n,), device="cuda", dtype=torch.int32)
vals = torch.arange(n, device="cuda", dtype=torch.int32)
want = torch.sort(keys.to(torch.int64), stable=True).indices.to(torch.int32)
k_in, v_in = keys.clone(), vals.clone()
k_alt, v_alt = torch.empty_like(keys), torch.empty_like(vals)
nbytes = int(_lib.lib.gsr_radix_scratch_bytes(n, bits))
scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
where = ctypes.c_int(0)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    rc = _lib.lib.gsr_radix_sort_pairs(n, bits, k_in.data_ptr(), k_alt.data_ptr(), v_in.data_ptr(), v_alt.data_ptr(), 0,
                                       scratch.data_ptr(), nbytes, ctypes.byref(where),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, _lib.last_error()
k_in.copy_(keys); v_in.copy_(vals)
graph.replay()
torch.cuda.synchronize()
got = v_alt if where.value else v_in
assert torch.equal(got, want), "the captured sort is wrong"
assert _lib.get_option(_lib.OPT_RADIX_RANK_ACTIVE) == 1, "outside the capture the device is tested and passes"
print("captured-ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "GSR_RADIX_RANK"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0 and "captured-ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_c2_full_frame_is_identical_with_either_rank():
    """BASELINE configs[1] at full size through every rank form: every public output and the sorted lists bit for bit
    (debug = True also runs the library's own (tile, depth, id) order check over the 3.5 M-entry list)."""
    import numpy as np
    from autovfx_amd import _lib, scenes
    from autovfx_amd.cameras import orbit_cameras
    from helpers import hip_forward_inference, hip_forward_raw
    cloud, cam = scenes.config_c2(), orbit_cameras(200, 960, 540)[100]
    outs = {}
    start = fallbacks()
    for mode in (0, 1):
        _lib.set_option(_lib.OPT_RADIX_RANK, mode)
        outs[mode] = (hip_forward_raw(cloud, cam, cull=True, debug=True), hip_forward_inference(cloud, cam, slabs=0, slab_first=40, debug=True))
    for k in ("color", "depth", "alpha", "radii", "depth_order", "point_list", "tile_keys", "ranges", "n_contrib"):
        np.testing.assert_array_equal(outs[0][0][k], outs[1][0][k], err_msg=k)
    for k in ("color", "depth", "alpha", "radii"):
        np.testing.assert_array_equal(outs[0][1][k], outs[1][1][k], err_msg="inference " + k)
        np.testing.assert_array_equal(outs[0][1][k], outs[0][0][k], err_msg="inference vs full " + k)
    assert outs[0][1]["slab_pairs"] == outs[1][1]["slab_pairs"] and len(outs[0][1]["slab_pairs"]) > 1
    assert fallbacks() == start, "a tile's LDS-add ranks failed the order check on this device"
    # ... and with an inversion injected into every wave of every sort of the frame (depth sort with dropped keys, tile sorts
    # that start at a folding scan, device-side pair counts): the order check has to catch them all and the repair to restore
    # the same lists
    _lib.set_option(_lib.OPT_RADIX_RANK, 3)
    hurt = (hip_forward_raw(cloud, cam, cull=True, debug=True), hip_forward_inference(cloud, cam, slabs=0, slab_first=40, debug=True))
    assert fallbacks() > start + 1000, "the injected inversions went unnoticed"
    for k in ("color", "depth", "alpha", "radii", "depth_order", "point_list", "tile_keys", "ranges", "n_contrib"):
        np.testing.assert_array_equal(hurt[0][k], outs[0][0][k], err_msg="repaired: " + k)
    for k in ("color", "depth", "alpha", "radii"):
        np.testing.assert_array_equal(hurt[1][k], outs[0][1][k], err_msg="repaired, inference: " + k)
