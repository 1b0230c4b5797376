#!/usr/bin/env python3
"""Golden vectors for the discrete Frechet distance: random curve pairs through the reference's frechet.distance.frechet_distance
(imported with the identity numba shim of tools/make_golden.py) -> tests/golden/frechet.npz."""
import importlib
import pathlib
import shutil
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import make_golden  # noqa: E402


def main():
    mods, cfg, tmp = make_golden.import_reference()
    fd = importlib.import_module("frechet.distance").frechet_distance
    rng = np.random.default_rng(99)
    out = {}
    K = 24
    dist = np.empty(K)
    for k in range(K):
        n, m = int(rng.integers(2, 15)), int(rng.integers(2, 15))
        if k < 6:
            m = n
        t_all = np.array([0.0, 0.5, 0.75, 1.0, 2.0, 4.0, 8.0, 16.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])
        ta = np.sort(rng.choice(t_all, n, replace=False)); tb = np.sort(rng.choice(t_all, m, replace=False))
        a = np.ascontiguousarray(np.stack([ta, rng.uniform(0.2, 3.0, n)], axis=1))
        b = np.ascontiguousarray(np.stack([tb, rng.uniform(0.2, 3.0, m)], axis=1))
        out[f"a{k}"] = a; out[f"b{k}"] = b
        dist[k] = fd(a, b)
    out["dist"] = dist
    np.savez_compressed(make_golden.OUT / "frechet.npz", **out)
    print(dist)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
