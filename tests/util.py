"""Helpers shared by the tests (golden loading, synthetic weights)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {k: z[k] for k in z.files}
    if "keys" in out:
        out["keys"] = [(k, tuple(s)) for k, s in json.loads(bytes(out["keys"]).decode())]
    return out


def golden_state_dict(name, seed=0):
    from oracle.weights import synth_state_dict
    g = load_golden(name)
    return synth_state_dict(g["keys"], seed), g
