"""Import-time constants of global_model/config.py that the network hot path reads.  MODEL: 0 distributive, 1 sequential,
2 combinatorial, 4 saturating (global_model/config.py:59-61; config.toml [global_model.models] default_model)."""
import os
import numpy as np

_NAMES = {"distributive": 0, "sequential": 1, "combinatorial": 2, "saturation": 4}
MODEL = _NAMES.get(os.environ.get("PHOSKIN_GLOBAL_MODEL", "distributive"), 4)
TIME_POINTS_PROTEIN = np.array([0.0, 0.5, 0.75, 1.0, 2.0, 4.0, 8.0, 16.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])
TIME_POINTS_RNA = np.array([4.0, 8.0, 15.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])
TIME_POINTS_PHOSPHO = TIME_POINTS_PROTEIN.copy()      # config.toml:366 phospho_protein
ODE_ABS_TOL = 1e-8      # config.toml:402-405
ODE_REL_TOL = 1e-8
ODE_MAX_STEPS = 200000
# Morris screening of the network (config.toml:347-354; read at global_model/config.py:134-139)
SENSITIVITY_PERTURBATION = 0.05
SENSITIVITY_TRAJECTORIES = 100
SENSITIVITY_LEVELS = 40
SENSITIVITY_TOP_CURVES = 20
SENSITIVITY_METRIC = "total_signal"
SEED = 42              # config.toml [global_model] seed (global_model/config.py)
LOSS_MODE = int(os.environ.get("PHOSKIN_LOSS_MODE", "0"))   # config.toml [global_model.loss] mode (global_model/config.py); 0 = squared error
# physical parameter bounds of the optimiser (config.toml:368-397 [global_model.bounds])
BOUNDS_CONFIG = {"c_k": (1e-3, 4.0), "A_i": (1e-6, 10.0), "B_i": (1e-3, 1.0), "C_i": (1e-3, 2.0), "D_i": (0.1, 0.5), "Dp_i": (0.05, 5.0),
                 "E_i": (1e-4, 10.0), "tf_scale": (2.0, 10.0)}
