"""Import-time constants of global_model/config.py that the network hot path reads.  MODEL: 0 distributive, 1 sequential,
2 combinatorial, 4 saturating (global_model/config.py:59-61; config.toml [global_model.models] default_model)."""
import os
import numpy as np

_NAMES = {"distributive": 0, "sequential": 1, "combinatorial": 2, "saturation": 4}
MODEL = _NAMES.get(os.environ.get("PHOSKIN_GLOBAL_MODEL", "distributive"), 4)
TIME_POINTS_PROTEIN = np.array([0.0, 0.5, 0.75, 1.0, 2.0, 4.0, 8.0, 16.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])
TIME_POINTS_RNA = np.array([4.0, 8.0, 15.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])
ODE_ABS_TOL = 1e-8      # config.toml:402-405
ODE_REL_TOL = 1e-8
ODE_MAX_STEPS = 200000
