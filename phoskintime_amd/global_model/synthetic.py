"""Seeded synthetic networks of the shape SURVEY.md section 8d prescribes for BASELINE configs 4 / 5 (the reference ships no data):
N proteins, 1-6 sites each, n_K kinases of which half are also proteins (driven), ~1.5 kinases per site with alpha ~ U(0.2, 1),
a signed sparse TF net, K(t) = 1 + 0.3 * smooth noise clipped at 1e-6, defaults as global_model/runner.py:515-524."""
from __future__ import annotations

import numpy as np


def make_network(N: int = 100, total_sites: int = 300, n_K: int = 40, n_tf_edges: int = 250, model: int = 0, seed: int = 20260519, max_sites: int | None = None):
    rng = np.random.default_rng(seed)
    # sites per protein: >= 0, sum = total_sites, <= 6 (model 2: <= 3)
    cap = max_sites if max_sites is not None else (3 if model == 2 else 6)
    n_sites = np.zeros(N, dtype=np.int32)
    while n_sites.sum() < total_sites:
        i = int(rng.integers(0, N))
        if n_sites[i] < cap:
            n_sites[i] += 1
    offset_s = np.concatenate([[0], np.cumsum(n_sites)[:-1]]).astype(np.int32)
    blk = (1 + (1 << n_sites.astype(np.int64))) if model == 2 else (2 + n_sites)
    offset_y = np.concatenate([[0], np.cumsum(blk)[:-1]]).astype(np.int32)
    # W: each site is hit by 1-2 kinases
    indptr = [0]; indices = []; data = []
    for _ in range(int(n_sites.sum())):
        ks = rng.choice(n_K, size=int(rng.integers(1, 3)), replace=False)
        for k in sorted(ks):
            indices.append(int(k)); data.append(float(rng.uniform(0.2, 1.0)))
        indptr.append(len(indices))
    # TF net (rows = targets), no self loops
    rows = [[] for _ in range(N)]
    edges = set()
    while len(edges) < n_tf_edges:
        a, b = (int(v) for v in rng.choice(N, size=2, replace=False))
        if (a, b) not in edges:
            edges.add((a, b)); rows[b].append((a, float(rng.uniform(-1, 1))))
    tptr = [0]; tind = []; tdat = []
    for r in rows:
        for a, w in sorted(r):
            tind.append(a); tdat.append(w)
        tptr.append(len(tind))
    tf_deg = np.array([sum(abs(w) for _, w in r) for r in rows]); tf_deg[tf_deg < 1e-12] = 1.0
    driver = np.full(N, -1, dtype=np.int32)
    kin_prot = rng.choice(N, size=n_K // 2, replace=False)
    driver[kin_prot] = np.arange(n_K // 2, dtype=np.int32)
    grid = np.array([0.0, 0.5, 0.75, 1.0, 2.0, 4.0, 8.0, 16.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])
    smooth = np.cumsum(rng.standard_normal((n_K, grid.size)), axis=1) / np.sqrt(np.arange(1, grid.size + 1))
    Kmat = np.maximum(1.0 + 0.3 * smooth, 1e-6)
    return dict(model=model, offset_y=offset_y, offset_s=offset_s, n_sites=n_sites,
                W_indptr=np.array(indptr, np.int32), W_indices=np.array(indices, np.int32), W_data=np.array(data),
                TF_indptr=np.array(tptr, np.int32), TF_indices=np.array(tind, np.int32), TF_data=np.array(tdat),
                tf_deg=tf_deg, driver_map=driver, kin_grid=grid, kin_Kmat=Kmat)


def default_candidate(net: dict) -> np.ndarray:
    """Physical defaults of runner.py:515-524 as one candidate row."""
    N = net["offset_y"].size; nK = net["kin_Kmat"].shape[0]; sites = int(net["n_sites"].sum())
    return np.concatenate([np.ones(nK), np.ones(N), np.full(N, 0.2), np.full(N, 0.5), np.full(N, 0.05), np.full(sites, 0.05), np.ones(N), [0.1]])


def random_candidates(net: dict, B: int, seed: int = 0, spread: float = 0.5) -> np.ndarray:
    """B physical candidates: defaults times log-normal factors (sigma = spread)."""
    rng = np.random.default_rng(seed)
    base = default_candidate(net)
    return base[None, :] * np.exp(spread * rng.standard_normal((B, base.size)))
