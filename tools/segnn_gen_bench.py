"""Forward time of SEGNN on the general-irreps kernels (csrc/lb_segnn_gen.hip) next to the fused lmax-1 kernels, DAM2D at
full size, one trajectory (BASELINE.json configs[4]'s graph).  Output: one line per configuration.

    python tools/segnn_gen_bench.py [--case dam2d] [--layers 10] [--reps 20]
"""
import argparse
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from lagrangebench_amd.case_setup import case_builder  # noqa: E402,F401
from lagrangebench_amd.data import make_case  # noqa: E402
from lagrangebench_amd.models import SEGNN, node_irreps  # noqa: E402
from tests._common import hip_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="dam2d")
    ap.add_argument("--layers", type=int, default=10)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", type=int, default=-1, help="index of the one configuration to run (for rocprofv3)")
    a = ap.parse_args()
    ds = make_case(a.case, n_trajs=1, extra_seq_length=2)
    ds.magnitude_features = True
    isl = ds.input_seq_length
    homog = bool(np.all(ds[0][1] == 0))
    irr = node_irreps(ds.metadata, isl, ds.external_force_fn is not None, True, homog)
    pos, pt = ds[0]
    hcase = hip_case(ds)
    feats, _ = hcase.allocate_eval((pos[None, :, :isl], pt[None]))
    eng = feats.engine
    st = eng.stats()
    print(f"case {a.case}: N = {pos.shape[0]}, E = {st.get('n_edges', st)}, layers {a.layers}")
    cfgs = [(1, 1, None), (1, 1, "instance"), (1, 1, "batch"), (2, 1, None), (1, 2, None), (2, 2, None), (2, 2, "batch")]
    for lh, la, norm in (cfgs if a.only < 0 else [cfgs[a.only]]):
        model = SEGNN(irr, "1x1o+1x0e", 64, lh, la, "1x1o", num_mp_steps=a.layers, n_vels=isl - 1, homogeneous_particles=homog, norm=norm)
        params = model.init_params(3)
        h = model.handle(eng, params)
        for _ in range(3):
            eng.segnn_forward(h)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(a.reps):
            acc = eng.segnn_forward(h)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / a.reps
        hid = "+".join(f"{model._hidden}x{l}{'eo'[l % 2]}" for l in range(lh + 1))
        print(f"lmax_hidden {lh} lmax_attributes {la} norm {str(norm):8s} hidden {hid:16s} path {'general' if model.generic else 'fused  '}"
              f"  {ms:8.3f} ms / forward   finite {bool(torch.isfinite(acc).all())}")
        h.close()


if __name__ == "__main__":
    main()
