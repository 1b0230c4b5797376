"""CPU: host-side logic of the drop-in layer that needs no GPU (engine cache lifetime, option plumbing)."""
import gc
import types

import numpy as np
import pytest


class _FakeEngine:
    made = 0
    closed = []

    def __init__(self, tag):
        self.tag = tag
        _FakeEngine.made += 1

    def close(self):
        _FakeEngine.closed.append(self.tag)


def test_engine_cache_is_keyed_by_live_object_and_evicts(monkeypatch):
    """engines were cached under id(sys) with no reference to sys, so a recycled id handed a NEW System the OLD topology and
    every System leaked its HBM.  Now: weak reference + identity check + finalizer that DROPS the cache entry.  ADVICE r2: the finalizer
    must not close engines -- a caller may still hold one; an engine frees its HBM when its own last reference dies."""
    from phoskintime_amd.global_model import simulate as gsim
    monkeypatch.setattr(gsim.NetworkEngine, "from_system", classmethod(lambda cls, sys, model, device=None: _FakeEngine((sys.name, model))))
    gsim._engines.clear(); _FakeEngine.made = 0; _FakeEngine.closed.clear()

    class Sys:                                   # weak-referenceable stand-in for global_model.network.System
        def __init__(self, name):
            self.name = name

    a = Sys("a")
    e0 = gsim.engine_for(a, 0)
    assert gsim.engine_for(a, 0) is e0 and _FakeEngine.made == 1            # cached
    e2 = gsim.engine_for(a, 2)
    assert e2 is not e0 and e2.tag == ("a", 2)                             # one engine per kinetic model
    key = id(a)
    del a; gc.collect()
    assert key not in gsim._engines and _FakeEngine.closed == []           # evicted with the System; e0 / e2 are still held and stay usable
    # a stale entry under a recycled id is never handed out: simulate it by planting a dead weakref under a live object's id
    b = Sys("b")
    dead = Sys("dead"); import weakref; ref = weakref.ref(dead); del dead; gc.collect()
    stale = _FakeEngine(("stale", 0))
    gsim._engines[id(b)] = (ref, {0: stale})
    eb = gsim.engine_for(b, 0)
    assert eb.tag == ("b", 0) and gsim._engines[id(b)][1] == {0: eb}
    # objects that cannot be weakly referenced: engines live on the object itself (SimpleNamespace), or a clear error (no __dict__)
    ns = types.SimpleNamespace(name="ns")
    e_ns = gsim.engine_for(ns, 0)
    assert gsim.engine_for(ns, 0) is e_ns and ns._pk_engines == {0: e_ns} and id(ns) not in gsim._engines
    with pytest.raises(TypeError):
        gsim.engine_for(5, 0)


def test_solver_opts_kernel_field_roundtrip(built_lib):
    from phoskintime_amd import _capi
    o = _capi.default_opts()
    assert o.kernel == _capi.KERNEL_AUTO and o.method == _capi.METHOD_LRP12
    assert _capi.default_opts(kernel="tpr").kernel == _capi.KERNEL_TPR and _capi.default_opts(kernel=1).kernel == _capi.KERNEL_GROUP
    with pytest.raises(KeyError):
        _capi.default_opts(kernel="warp")


_REF_SCRIPT = r'''
import sys, pathlib, numpy as np, pandas as pd
repo = pathlib.Path(sys.argv[1]); out = sys.argv[2]; model_name = sys.argv[3]
sys.path.insert(0, str(repo / "tools"))
import make_golden_network as mg
mods, tmp = mg.import_reference(model_name)
cfg, net, bm = mods["config"], mods["network"], mods["buildmat"]
rng = np.random.default_rng(31 + 100 * cfg.MODEL)                      # the network of tools/make_golden_pins.py main_network
prots, kinases, inter, tf_net = mg.synth_frames(rng, 8, 3, 2, 2, 10)
idx = net.Index(inter, tf_interactions=tf_net, kin_beta_map={k: float(rng.uniform(0.5, 1.5)) for k in kinases}, tf_beta_map={})
grid = np.asarray(cfg.TIME_POINTS_PROTEIN, float)
rows = []
for k in idx.kinases:
    base = 1.0 + 0.5 * np.sin(rng.uniform(0, 6) + np.arange(grid.size) * rng.uniform(0.2, 0.8))
    for t, v in zip(grid, base):
        rows.append(dict(protein=k, time=float(t), fc=float(max(v, 1e-6))))
kin_in = net.KinaseInput(idx.kinases, pd.DataFrame(rows))
W = bm.build_W_parallel(inter, idx, n_cores=1)
tf_mat = bm.build_tf_matrix(tf_net, idx, tf_beta_map={}, kin_beta_map={})
tf_deg = np.asarray(np.abs(tf_mat).sum(axis=1)).ravel().astype(np.float64); tf_deg[tf_deg < 1e-12] = 1.0
d = dict(c_k=np.ones(len(idx.kinases)), A_i=np.ones(idx.N), B_i=np.full(idx.N, 0.2), C_i=np.full(idx.N, 0.5), D_i=np.full(idx.N, 0.05),
         Dp_i=np.full(idx.total_sites, 0.05), E_i=np.ones(idx.N), tf_scale=0.1)
sysm = net.System(idx, W, tf_mat, kin_in, d, tf_deg)
sys.path.insert(0, str(repo))
from phoskintime_amd.global_model.engine import NetworkEngine
np.savez(out, **NetworkEngine.desc_from_system(sysm))
'''


@pytest.mark.skipif(not __import__("pathlib").Path("/root/reference").exists(), reason="needs the reference tree (build container only)")
@pytest.mark.parametrize("model_name,m", [("distributive", 0), ("combinatorial", 2)])
def test_from_system_packs_the_real_reference_System(tmp_path, model_name, m):
    """a23 was only ever tested with a SimpleNamespace.  Here the REAL global_model.network.System (imported in a child
    process: the reference fixes MODEL at import and wants its own CWD) goes through NetworkEngine.desc_from_system, and the result must
    equal what the reference itself packs into odeint_args (stored in the pins fixture made from the same seeded network)."""
    import subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    script = tmp_path / "pack.py"; script.write_text(_REF_SCRIPT)
    out = tmp_path / "desc.npz"
    r = subprocess.run([sys.executable, str(script), str(root), str(out), model_name], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out); want = np.load(root / "tests" / "golden" / f"pins_network_m{m}.npz")
    for k in ("offset_y", "offset_s", "n_sites", "W_indptr", "W_indices", "W_data", "TF_indptr", "TF_indices", "TF_data", "tf_deg", "driver_map",
              "kin_grid", "kin_Kmat"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    assert got["driver_map"].dtype == np.int32 and (got["driver_map"] >= 0).sum() >= 2
