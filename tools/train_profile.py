"""Training-step profile helper: `rocprofv3 --kernel-trace --stats -- python tools/train_profile.py [workload] [steps]`
runs a few GNS-10-128 training steps (bench.py's train_step_lines workload) so that the per-kernel times can be read."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from lagrangebench_amd.data import make_case  # noqa: E402
from lagrangebench_amd.models import GNS  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "tgv3d"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
device = torch.device("cuda:0")
ds = make_case(workload, n_trajs=1, extra_seq_length=2)
dim, isl = len(ds.box), ds.input_seq_length
if len(sys.argv) > 3 and sys.argv[3] == "segnn":   # python tools/train_profile.py dam2d 10 segnn
    from lagrangebench_amd.models import SEGNN, node_irreps
    ds.magnitude_features = True
    homog = bool((ds[0][1] == 0).all())
    model = SEGNN(node_irreps(ds.metadata, isl, ds.external_force_fn is not None, True, homog), "1x1o+1x0e", 64, 1, 1, "1x1o",
                  num_mp_steps=10, n_vels=isl - 1, homogeneous_particles=homog)
    params = model.init_params(1234)
else:
    model = GNS(dim, bench.D, 2, 10, 16)
    node_in, edge_in = bench.gns_widths(ds)
    params = model.init_params(1234, node_in, edge_in, decoder_scale=1.0)
case = bench.hip_case(ds)
pos, pt = ds[0]
feats, _ = case.allocate_eval((pos[None, :, :isl], pt[None]))
eng = feats.engine
th = model.train_handle(eng, params)
target = torch.randn((1, len(pt), dim), generator=torch.Generator().manual_seed(5)).to(device)
for _ in range(2):
    th.zero_grad(); th.loss_grad(target, 1.0); th.adamw_step(1e-4)
torch.cuda.synchronize(device)
# three timed groups of K steps: the first one also warms the clocks up (a process that starts on an idle GPU reads ~10 %
# slower over its first 20 steps); the line reports the last group, the others follow in brackets
times = []
for rep in range(3 if K >= 10 else 1):
    t0 = time.perf_counter()
    for _ in range(K):
        th.zero_grad(); loss = th.loss_grad(target, 1.0); th.adamw_step(1e-4)
    torch.cuda.synchronize(device)
    times.append(1e3 * (time.perf_counter() - t0) / K)
print(workload, "E", eng.stats()["n_edges_total"], "ms/step", times[-1], "loss", float(loss), "repeated steps", th.math_fallbacks(),
      "groups", [round(x, 3) for x in times])
