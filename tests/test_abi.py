"""CPU checks of the drop-in boundary: the C-ABI library loads (no GPU needed) and exports every
symbol include/lbhip.h declares; the ctypes table binds exactly that set; argument validation
that does not touch the device works."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "lbhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lb_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from lagrangebench_amd import build
    build.build()  # hipcc cross-compiles gfx950 without a GPU
    from lagrangebench_amd import _lib
    return _lib.load()


def test_header_and_ctypes_table_agree(lib):
    from lagrangebench_amd import _lib
    names = _header_functions()
    assert len(names) >= 30
    assert sorted(_lib._SIGS) == names
    for n in names:
        assert hasattr(lib, n), f"liblbhip.so does not export {n}"


def test_version_and_error_strings(lib):
    assert lib.lb_version() >= 100
    assert lib.lb_strerror(0) == b"ok"
    assert b"argument" in lib.lb_strerror(-1)
    assert lib.lb_timer_count() >= 10
    names = [lib.lb_timer_name(i).decode() for i in range(lib.lb_timer_count())]
    assert "edge_mlp" in names and "aggregate" in names and "neighbors" in names


def test_argument_validation_without_device(lib):
    from lagrangebench_amd._lib import CaseDesc
    h = C.c_void_p()
    assert lib.lb_engine_create(None, None, C.byref(h)) == -1
    d = CaseDesc()
    d.dim, d.n_particles, d.batch, d.isl, d.r_cutoff = 4, 10, 1, 6, 0.1
    assert lib.lb_engine_create(C.byref(d), None, C.byref(h)) == -1
    assert b"dim" in lib.lb_last_error()
    d.dim, d.isl = 2, 1
    assert lib.lb_engine_create(C.byref(d), None, C.byref(h)) == -1
    assert lib.lb_nl_update(None) == -1


def test_engine_refuses_cpu():
    """The product path must fail loudly without a HIP device - no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lagrangebench_amd._lib import LbHipError
    from lagrangebench_amd.data import make_case
    from tests._common import hip_case
    ds = make_case("small2d", n_trajs=1, extra_seq_length=2)
    case = hip_case(ds)
    with pytest.raises(LbHipError):
        case.allocate_eval((ds[0][0][:, :6], ds[0][1]))


def test_product_code_never_imports_the_oracle():
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "lagrangebench_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "lb_oracle" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
