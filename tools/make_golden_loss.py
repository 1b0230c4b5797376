#!/usr/bin/env python3
"""Golden vectors for global_model.lossfn (all 8 LOSS_MODEs, both state layouts) and the objective assembly of
optproblem.GlobalODE_MOO._evaluate, made by running the reference's loss functions on the trajectories already stored in
tests/golden/network_m{0,2}_small.npz (build container only; same import recipe as tools/make_golden_network.py)."""
import sys, pathlib, importlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import make_golden_network as mg

OUT = mg.OUT


def main():
    mods, tmp = mg.import_reference("distributive")
    lossfn = importlib.import_module("global_model.lossfn")
    for m, fn in ((0, lossfn.loss_function_noncomb), (2, lossfn.loss_function_comb)):
        g = np.load(OUT / f"network_m{m}_small.npz")
        N, T = int(g["N"]), g["t_eval"].size
        rng = np.random.default_rng(5 + m)
        n_sites = g["n_sites"]
        second = (1 << n_sites.astype(np.int64)) if m == 2 else n_sites          # prepare_fast_loss_data: prot_map[:, 1] = n_states or n_sites
        prot_map = np.stack([g["offset_y"], second], axis=1).astype(np.int32)
        n_prot, n_rna, n_pho = 40, 30, 35
        with_sites = np.where(n_sites > 0)[0]
        d = dict(p_prot=rng.integers(0, N, n_prot).astype(np.int32), t_prot=rng.integers(0, T, n_prot).astype(np.int32),
                 obs_prot=rng.uniform(0.3, 3.0, n_prot), w_prot=rng.uniform(0.5, 2.0, n_prot),
                 p_rna=rng.integers(0, N, n_rna).astype(np.int32), t_rna=rng.integers(0, T, n_rna).astype(np.int32),
                 obs_rna=rng.uniform(0.3, 3.0, n_rna), w_rna=rng.uniform(0.5, 2.0, n_rna))
        pp = rng.choice(with_sites, n_pho).astype(np.int32)
        d.update(p_pho=pp, s_pho=np.array([rng.integers(0, n_sites[i]) for i in pp], dtype=np.int32), t_pho=rng.integers(0, T, n_pho).astype(np.int32),
                 obs_pho=rng.uniform(0.1, 4.0, n_pho), w_pho=rng.uniform(0.5, 2.0, n_pho))
        base = dict(prot_base_idx=0, rna_base_idx=int(np.where(g["t_eval"] == 4.0)[0][0]), pho_base_idx=0)
        K = g["Y_lsoda8"].shape[0]
        L = np.empty((8, K, 3))
        for mode in range(8):
            lossfn.LOSS_MODE = mode                   # import-time constant of the reference (lossfn.py:22), read at call time un-jitted
            for k in range(K):
                L[mode, k] = fn(np.ascontiguousarray(g["Y_lsoda8"][k]), d["p_prot"], d["t_prot"], d["obs_prot"], d["w_prot"],
                                d["p_rna"], d["t_rna"], d["obs_rna"], d["w_rna"], d["p_pho"], d["s_pho"], d["t_pho"], d["obs_pho"], d["w_pho"],
                                prot_map, base["prot_base_idx"], base["rna_base_idx"], base["pho_base_idx"])
        np.savez_compressed(OUT / f"network_loss_m{m}.npz", model=m, prot_map=prot_map, loss_sums=L, Y=g["Y_lsoda8"], **d, **base)
        print("wrote loss golden for model", m, L[0, 0], flush=True)


if __name__ == "__main__":
    main()
