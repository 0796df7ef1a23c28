"""Regenerate the committed golden fixtures from the reference's own test DATA.

Run in the build container only (needs /root/reference and /opt/conda/bin/h5dump):

    python tests/golden/make_golden.py

What it produces (data only - no reference source text is copied):

* ``lj3d_{valid,test}.npz``  - the Lennard-Jones debugging dataset the reference's
  ``tests/rollout_test.py`` / ``tests/runner_test.py`` run on
  (``/root/reference/tests/3D_LJ_3_1214every1/{valid,test}.h5``, decoded with
  ``h5dump -b`` because h5py is not installed) plus its ``metadata.json``.
* ``tgv2d_metadata.json`` - the metadata block of the published 2D_TGV_2500_10kevery100 dataset as
  PRINTED in the output of ``/root/reference/notebooks/datasets.ipynb`` (cell ``dataset.metadata``)
  plus the neighbor-list capacities the reference logged for that dataset in
  ``notebooks/tutorial.ipynb`` ("From (2, 21057) to ..."): cell OUTPUTS (data), not source.
* ``case_test_vectors.json`` is NOT generated - it is a hand transcription of the
  numeric literals in ``/root/reference/tests/case_test.py`` (inputs :14-65, expected
  outputs :79,89,97,105-111,116-128,198) and is committed as-is.
"""
import json
import os
import subprocess
import tempfile

import numpy as np

REF = "/root/reference/tests/3D_LJ_3_1214every1"
H5DUMP = "/opt/conda/bin/h5dump"
HERE = os.path.dirname(os.path.abspath(__file__))


def dump(h5, dset, dtype, shape):
    with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
        subprocess.run(
            [H5DUMP, "-d", dset, "-b", "LE", "-o", tmp.name, h5],
            check=True,
            stdout=subprocess.DEVNULL,
        )
        return np.fromfile(tmp.name, dtype=dtype).reshape(shape)


def main():
    with open(os.path.join(REF, "metadata.json")) as f:
        meta = json.load(f)
    for split in ["valid", "test"]:
        h5 = os.path.join(REF, f"{split}.h5")
        T = meta["sequence_length_test"]
        pos = dump(h5, "/00000/position", "<f4", (T, 3, 3))
        ptype = dump(h5, "/00000/particle_type", "<i4", (3,))
        np.savez(os.path.join(HERE, f"lj3d_{split}.npz"), position=pos, particle_type=ptype)
        print(split, pos.shape, pos.dtype, ptype, float(pos.min()), float(pos.max()))
    with open(os.path.join(HERE, "lj3d_metadata.json"), "w") as f:
        json.dump(meta, f, indent=1)


def notebook_fixture():
    import ast
    import re
    nb = json.load(open("/root/reference/notebooks/datasets.ipynb"))
    md = None
    for c in nb["cells"]:
        if c["cell_type"] == "code" and "".join(c["source"]).strip() == "dataset.metadata":
            txt = "".join("".join(o.get("data", {}).get("text/plain", [])) for o in c["outputs"])
            md = ast.literal_eval(txt)
    assert md is not None and md["num_particles_max"] == 2500
    tut = json.load(open("/root/reference/notebooks/tutorial.ipynb"))
    caps = []
    for c in tut["cells"]:
        if c["cell_type"] == "code":
            for o in c.get("outputs", []):
                caps += [[int(a), int(b)] for a, b in re.findall(r"From \(2, (\d+)\) to \(2, (\d+)\)", "".join(o.get("text", [])))]
    out = {"metadata": md, "logged_capacity_changes": caps, "capacity_multiplier": 1.25,
           "source": "notebooks/datasets.ipynb (output of cell `dataset.metadata`), notebooks/tutorial.ipynb (trainer / infer logs)"}
    with open(os.path.join(HERE, "tgv2d_metadata.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("tgv2d metadata:", md["default_connectivity_radius"], "capacities:", caps[:3], "...")


if __name__ == "__main__":
    main()
    notebook_fixture()
