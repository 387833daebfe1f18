import os, sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import ROOT, DeviceSim, load_hip
import test_oracle_golden as tg
g = dict(np.load(os.path.join(ROOT, "tests/golden/go2_heightfield_sequence.npz")))
hip = load_hip()
s = tg._mk(hip, g, sim=DeviceSim)
for t in tg.run_sequence(s, hip, g, None):
    s.torch.cuda.synchronize()
    for k, want in (("privileged_obs_buf", g["priv"][t]), ("measured_heights", g["measured_heights"][t]), ("obs_buf", g["obs"][t]), ("rew_buf", g["rew"][t]), ("terrain_levels", g["terrain_levels"][t])):
        got = np.asarray(getattr(s, k))
        bad = np.argwhere(np.abs(got - want) > 1e-4)
        if len(bad):
            print("t", t, k, "nbad", len(bad), "envs", np.unique(bad[:, 0]) if bad.ndim > 1 else bad[:8].ravel(), "cols", (np.unique(bad[:, 1])[:40] if bad.ndim > 1 and bad.shape[1] > 1 else ""))
            if k == "privileged_obs_buf":
                e, c = bad[0]; print("  e.g.", e, c, got[e, c], want[e, c], "reset", g["reset"][t][e], "root_in", g["root_in"][t][e][:3], "root_out", g["root_out"][t][e][:3])
    if t > 6: break
