import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The round-end run is `pytest tests -x -q -m gpu`: with -x one failure hides everything collected behind it.  Files run in the
# order below -- the forward path the benchmark times first, the sort, the raw-parameter path, the C ABI, the adjacent
# components, RCCL, and the gradient tests (sums of float atomics against a truth-based bar) and the comparisons with the
# reference's own kernels last -- so that a red run reports as much as it can.  Inside a file the order is the file's own.
_FILE_ORDER = ("test_parity_gpu", "test_round3_options_gpu", "test_poison_gpu", "test_golden", "test_oracle_kats", "test_oracle_ref_pin",
               "test_radix_gpu", "test_raw_gpu", "test_render_mirror", "test_install_hook", "test_api_compat", "test_cabi",
               "test_cabi_native", "test_host_logic", "test_compositor", "test_dynamic_scene", "test_frame_io", "test_png_gpu",
               "test_resize_gpu", "test_rccl_gpu", "test_frame_parallel", "test_bench_launch", "test_oracle_backward", "test_oracle_truth",
               "test_backward_gpu", "test_raw_autograd_gpu", "test_reference_hip_gpu")


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_FILE_ORDER)}
    key = lambda item: rank.get(os.path.splitext(os.path.basename(str(item.fspath)))[0], len(_FILE_ORDER) // 2)
    items.sort(key=key)   # stable: the order inside a file stays


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import cpu_oracle
    cpu_oracle.build()


@pytest.fixture(autouse=True)
def _reset_binding_state():
    """Process-wide switches of the binding a test may have flipped go back to their defaults after every test: the opt-in
    geometry reuse between two rasterizer calls and the allocation-poison test hook."""
    yield
    mod = sys.modules.get("diff_gaussian_rasterization._C")
    if mod is not None:
        mod.set_geometry_cache(None)
        mod.set_alloc_poison(None)
    lib = sys.modules.get("autovfx_amd._lib")
    if lib is not None:   # library options a failing test may have left behind: one red test must not colour the ones after it
        for opt, value in ((lib.OPT_TILE_CULL, 1), (lib.OPT_SLABS, 2), (lib.OPT_SLAB_FIRST, 400), (lib.OPT_DEFER_COLOUR, 1),
                           (lib.OPT_SLAB_MIN_REST, 3000000), (lib.OPT_BACKWARD_DETERMINISTIC, 0), (lib.OPT_GRAD_SLABS, 1)):
            lib.set_option(opt, value)
