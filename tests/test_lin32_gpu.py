"""The training step's fp32-MFMA GEMMs (csrc/lb_lin32.h: k_pack_w, k_lin32, k_lin32f with every epilogue) against an fp64
reference, kernel by kernel.  The gradient tests of tests/test_train.py see these kernels only through whole-network sums;
here each (shape, epilogue) pair is launched on its own by tools/lin_bench.hip - compiled in place with hipcc, as
lagrangebench_amd/build.py compiles the library - and 400 sampled rows (the first and last 40 among them) of every output are
compared with a double-precision evaluation of  Y = X W / dX = dY W^T  + bias / ReLU / ReLU mask / += / LayerNorm + residual /
two gathered rows: relative error <= 2e-6 of sum |x w| + 1 (fp32 accumulation of 128 - 256 terms is ~2e-7).
The CPU part checks the fragment order k_pack_w writes against the MFMA operand layout the kernels assume.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fragment_order_is_a_permutation_of_the_operand():
    """k_pack_w: entry ((j * NOB + mb) * 64 + lane) * 4 + i holds Wop[16 j + 4 (lane >> 4) + i][16 mb + (lane & 15)] - every
    element of a (NR x NO) operand appears exactly once, padding is zero (host restatement of the index map)."""
    for NR, NO, NOB in [(128, 128, 8), (256, 128, 8), (32, 128, 8), (3, 128, 8), (128, 3, 1)]:
        NJ = (NR + 15) // 16
        W = np.arange(1, NR * NO + 1, dtype=np.float64).reshape(NR, NO)
        out = np.zeros(NJ * NOB * 256)
        idx = np.arange(out.size)
        i, ln, q = idx & 3, (idx >> 2) & 63, idx >> 8
        mb, j = q % NOB, q // NOB
        k, m = 16 * j + 4 * (ln >> 4) + i, 16 * mb + (ln & 15)
        ok = (k < NR) & (m < NO)
        out[ok] = W[k[ok], m[ok]]
        assert np.count_nonzero(out) == NR * NO and np.array_equal(np.sort(out[out > 0]), np.sort(W.ravel()))


@pytest.mark.gpu
def test_lin32_kernels_match_fp64(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "lin_bench")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "lagrangebench_amd", "csrc"),
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "lin_bench.hip"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # 20011 edge-sized rows (not a multiple of 16), 1777 node-sized rows: seconds, every kernel and epilogue
    r = subprocess.run([exe, "20011", "1777"], capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:])
    assert r.returncode == 0 and "all results match" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
    assert "WRONG" not in r.stdout
