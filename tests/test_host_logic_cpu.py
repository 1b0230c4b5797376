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
    every System leaked its HBM.  Now: weak reference + identity check + finalizer that closes the engines."""
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
    assert key not in gsim._engines and sorted(_FakeEngine.closed) == [("a", 0), ("a", 2)]      # evicted and closed with the System
    # a stale entry under a recycled id is never handed out: simulate it by planting a dead weakref under a live object's id
    b = Sys("b")
    dead = Sys("dead"); import weakref; ref = weakref.ref(dead); del dead; gc.collect()
    stale = _FakeEngine(("stale", 0))
    gsim._engines[id(b)] = (ref, {0: stale})
    eb = gsim.engine_for(b, 0)
    assert eb.tag == ("b", 0) and ("stale", 0) in _FakeEngine.closed
    with pytest.raises(TypeError):
        gsim.engine_for(types.SimpleNamespace.__call__.__self__ if False else 5, 0)          # ints cannot be weakly referenced


def test_solver_opts_kernel_field_roundtrip(built_lib):
    from phoskintime_amd import _capi
    o = _capi.default_opts()
    assert o.kernel == _capi.KERNEL_AUTO and o.method == _capi.METHOD_LRP12
    assert _capi.default_opts(kernel="tpr").kernel == _capi.KERNEL_TPR and _capi.default_opts(kernel=1).kernel == _capi.KERNEL_GROUP
    with pytest.raises(KeyError):
        _capi.default_opts(kernel="warp")
