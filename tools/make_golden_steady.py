#!/usr/bin/env python3
"""Golden vectors for the initial-condition helpers: runs the reference's steady.init{dist,succ,rand}.initial_condition(n) (SLSQP) in
this container and stores the returned lists.  Same import arrangement as tools/make_golden.py (writable temp copy, identity numba shim).

  python tools/make_golden_steady.py      ->  tests/golden/steady_init.npz
"""
import importlib
import pathlib
import shutil
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import make_golden  # noqa: E402


def main():
    mods, cfg, tmp = make_golden.import_reference()
    out = {}
    plan = {"initdist": (1, 2, 3, 4, 8, 14, 30), "initsucc": (1, 2, 4, 14), "initrand": (1, 2, 3, 4, 5)}
    for name, ns in plan.items():
        m = importlib.import_module(f"steady.{name}")
        for n in ns:
            y = np.asarray(m.initial_condition(n), float)
            out[f"{name}_n{n}"] = y
            print(name, n, y[:6], flush=True)
    np.savez_compressed(make_golden.OUT / "steady_init.npz", **out)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
