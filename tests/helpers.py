"""Fixture loading + comparison helpers shared by the test modules."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        self.meta = json.loads(str(z["meta"]))
        self.groups = {}
        for k in z.files:
            if k == "meta":
                continue
            g, n = k.split("/", 1)
            self.groups.setdefault(g, {})[n] = torch.from_numpy(np.array(z[k]))

    def __getattr__(self, g):
        try:
            return self.__dict__["groups"][g]
        except KeyError:
            raise AttributeError(g)

    @property
    def inp(self):
        return self.groups["in"]


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b| -- the 'relative fp32' measure BASELINE.json's 1e-3 tolerance is stated in."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    denom = b.abs().max().item()
    return (a - b).abs().max().item() / (denom if denom > 0 else 1.0)


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
    return e
