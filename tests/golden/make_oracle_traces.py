"""Freezes what the CPU oracle (oracle/oracle.c) produces TODAY on two seeded scenes, so that a later change to
the oracle -- the checker every GPU parity test trusts -- cannot drift unnoticed:
  st20  the reference's own bundle-adjustment size (29 cameras x 600 landmarks, scenes.st20_scene())
  c2    BASELINE config C2 (2 cameras, 5 000 landmarks, scenes.two_view_scene())
for each: the LM cost trace, the accept/reject sequence, the iteration count and the final parameters (all cameras,
the first 20 landmarks).  Run from the repository root:  python tests/golden/make_oracle_traces.py

This does NOT pin the oracle against the reference: the reference cannot be built here (Ceres, Sophus, Eigen are
absent) and holds no vectors for this leg -- "parity unpinned"  stands.  It pins the oracle
against ITSELF, i.e. it is a regression fixture."""
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(name, s, O):
    o = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    summ, tr = o.solve()
    return {"scene": name, "n_cams": int(o.nc), "n_pts": int(o.np_), "n_obs": int(o.no),
            "num_iterations": int(summ.num_iterations), "termination_type": int(summ.termination_type),
            "initial_cost": float(summ.initial_cost), "final_cost": float(summ.final_cost),
            "cost_trace": [float(x) for x in tr[:, 0]], "accepted": [int(x) for x in tr[:, 6]],
            "radius_trace": [float(x) for x in tr[:, 5]],
            "final_cams": o.cams.reshape(-1).tolist(), "final_pts_head": o.pts[:20].reshape(-1).tolist()}


if __name__ == "__main__":
    import oracle_py as O
    O.build()
    scenes = importlib.import_module("slam-tricks_amd.scenes")
    out = {"generator": "tests/golden/make_oracle_traces.py", "oracle": "oracle/oracle.c (self-regression, not a reference pin)",
           "st20": run("st20_scene()", scenes.st20_scene(), O),
           "c2": run("two_view_scene(n_pts=5000)", scenes.two_view_scene(n_pts=5000), O)}
    with open(os.path.join(HERE, "oracle_traces.json"), "w") as f:
        json.dump(out, f, indent=0)
    print({k: (v["num_iterations"], v["final_cost"]) for k, v in out.items() if isinstance(v, dict)})
