"""CPU known-answer tests for the integrator coefficient tables compiled into the kernels.

Every table in csrc/ was typed in from the literature or derived offline; these tests re-derive / re-verify them in multi-precision
arithmetic (mpmath) and compare with the literals parsed out of the HIP sources, so a typo in a constant cannot survive:
  * RODAS4 (Hairer-Wanner)      : the 8 Rosenbrock order-4 conditions            (tools/check_rodas4.py)
  * ROS34PW2 (Rang-Angermann)   : the 8 order-3 W-method conditions, stiff accuracy, R(inf) = 0   (tools/check_ros34pw2.py)
  * RODAS4 resolvent weights    : re-derived from the stage table                 (tools/rodas4_resolvent.py)
  * LRP8 weights                : order 7 / embedded order 6 conditions solved afresh, A-stability on the imaginary axis (tools/restricted_pade.py)
"""
import importlib.util
import re
import sys
from pathlib import Path

import mpmath as mp
import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "phoskintime_amd" / "csrc"


def _load(name):
    spec = importlib.util.spec_from_file_location(name, ROOT / "tools" / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _literals(text, name):
    """Numbers of `constexpr double NAME... = {a, b, ...}` or `NAME = value` in a source file."""
    m = re.search(name + r"\s*(?:\[\d*\])?\s*=\s*\{([^}]*)\}", text)
    if m:
        return [mp.mpf(x.strip()) for x in m.group(1).split(",") if x.strip()]
    m = re.search(r"\b" + name + r"\s*=\s*([-+0-9.eE]+)", text)
    assert m, name
    return [mp.mpf(m.group(1))]


def test_rodas4_table_satisfies_order_conditions_and_matches_source():
    mp.mp.dps = 50
    src = (CSRC / "pk_solve_kernel.hpp").read_text()
    chk = _load("check_rodas4")
    res = chk.conds(chk.b)
    assert max(abs(v) for v in res.values()) < mp.mpf("1e-14")
    emb = chk.conds(chk.bh)
    assert max(abs(emb[k]) for k in ("1", "2", "3a", "3b")) < mp.mpf("1e-14") and abs(emb["4a"]) > 1e-3      # embedded: order 3 only
    for (i, j), v in chk.a.items():
        assert _literals(src, f"A{i}{j}")[0] == mp.mpf(v)
    for (i, j), v in chk.c.items():
        assert _literals(src, f"C{i}{j}")[0] == mp.mpf(v)


def test_rodas4_resolvent_weights_match_source():
    mp.mp.dps = 60
    src = (CSRC / "pk_solve_kernel.hpp").read_text()
    rr = _load("rodas4_resolvent")
    for k in range(6):
        assert abs(_literals(src, f"RB{k + 1}")[0] - rr.beta[k]) < mp.mpf("1e-16")
        if k >= 1:
            assert abs(_literals(src, f"RE{k + 1}")[0] - rr.eps[k]) < mp.mpf("1e-16")
    assert abs(rr.eps[0]) < mp.mpf("1e-30")
    # the resolvent form reproduces exp(z) to order 4 and is L-stable
    z = mp.mpf("0.01")
    R = 1 + sum(rr.beta[k] * z / (1 - rr.g * z) ** (k + 1) for k in range(6))
    assert abs(R - mp.e ** z) < 10 * z ** 5
    zi = mp.mpf("-1e9")
    assert abs(1 + sum(rr.beta[k] * zi / (1 - rr.g * zi) ** (k + 1) for k in range(6))) < mp.mpf("1e-7")


@pytest.mark.parametrize("name,s_,gam", [("PK_METHOD_LRP8", 8, "0.22"), ("PK_METHOD_LRP12", 12, "0.16")])
def test_lrp_weights_rederived_and_a_stable(name, s_, gam):
    mp.mp.dps = 60
    src = (CSRC / "pk_solve_kernel.hpp").read_text()
    rp = _load("restricted_pade")
    blk = src[src.index("struct ResolventTab<%s>" % name):]
    blk = blk[:blk.index("\n};")]
    g = _literals(blk, "GAM")[0]
    assert g == mp.mpf(gam)
    B = _literals(blk, "B"); E = _literals(blk, "E")
    assert len(B) == s_ and len(E) == s_ and _literals(blk, "NS")[0] == s_ and _literals(blk, "Q")[0] == s_ - 1
    beta = rp.solve_weights(s_, g, s_ - 1)
    bh = rp.solve_weights(s_, g, s_ - 2, extra_zero=(s_,))
    for k in range(s_):
        assert abs(B[k] - beta[k]) < mp.mpf("2e-15") * max(1, abs(beta[k]))
        assert abs(E[k] - (beta[k] - bh[k])) < mp.mpf("2e-15") * max(1, abs(beta[k]))
    # order: R(z) - exp(z) = O(z^s) ; embedded O(z^(s-1))
    z = mp.mpf("0.05")
    assert abs(rp.R(beta, g, z) - mp.e ** z) < 1e-3 * z ** s_ * 100
    assert abs(rp.R(bh, g, z) - mp.e ** z) < 1e-2 * z ** (s_ - 1) * 100
    # L-stability and A-stability (poles sit at 1/gamma > 0; on the imaginary axis |R| <= 1)
    assert abs(rp.R(beta, g, mp.mpf("-1e12"))) < mp.mpf("1e-10") and abs(rp.R(bh, g, mp.mpf("-1e12"))) < mp.mpf("1e-10")
    # LRP8: strictly A-stable.  LRP12: the embedded method strictly, the propagated one up to its own leading error term
    # (|R(iy)|^2 - 1 ~ 2 C_12 y^12 with C_12 > 0 near |y| ~ 1): |R(iy)| <= 1 + 4.5e-9, as the source comment states
    assert rp.a_stable(beta, g, n=1500) <= 1 + (mp.mpf("1e-20") if s_ == 8 else mp.mpf("4.6e-9"))
    assert rp.a_stable(bh, g, n=1500) <= 1 + mp.mpf("1e-20")
    if s_ == 12:
        assert all(abs(rp.R(beta, g, mp.mpc(0, y))) < 1 for y in (2, 3, 5, 10, 100, 1e4))
        assert all(abs(rp.R(beta, g, mp.mpf(-x))) <= 1 for x in (1e-3, 0.1, 1, 3, 10, 100, 1e6))       # the negative real axis
    # the scalar-C oracle carries the same tables
    csrc = (CSRC.parents[1] / "oracle" / "lrp8_dist.c").read_text()
    cb = _literals(csrc, "LB%d" % s_); ce = _literals(csrc, "LE%d" % s_)
    assert [float(x) for x in cb] == [float(x) for x in B] and [float(x) for x in ce] == [float(x) for x in E]


def test_ros34pw2_table_satisfies_w_conditions_and_matches_source():
    mp.mp.dps = 50
    src = (CSRC / "pk_network_solve.hpp").read_text()
    chk = _load("check_ros34pw2")
    assert max(abs(v) for v in chk.conds(chk.b).values()) < mp.mpf("1e-14")
    emb = chk.conds(chk.bh)
    assert max(abs(emb[k]) for k in ("b.1 = 1", "b.A1 = 1/2", "b.G1 = 0")) < mp.mpf("1e-14")
    assert abs(1 - (chk.b * (chk.B ** -1) * chk.one)[0]) < mp.mpf("1e-14")            # R(inf) = 0
    names = {"A21": chk.a[1, 0], "A31": chk.a[2, 0], "A32": chk.a[2, 1], "A41": chk.a[3, 0], "A42": chk.a[3, 1], "A43": chk.a[3, 2],
             "C21": chk.c[1, 0], "C31": chk.c[2, 0], "C32": chk.c[2, 1], "C41": chk.c[3, 0], "C42": chk.c[3, 1], "C43": chk.c[3, 2],
             "E1": chk.m[0] - chk.mh[0], "E2": chk.m[1] - chk.mh[1], "E3": chk.m[2] - chk.mh[2], "E4": chk.m[3] - chk.mh[3], "GAM": chk.g}
    ns = src[src.index("namespace rosw"):src.index("}  // namespace rosw")]
    for k, v in names.items():
        assert abs(_literals(ns, k)[0] - v) < mp.mpf("1e-15") * max(1, abs(v)), k
    assert abs(chk.m[3] - 1) < mp.mpf("1e-15") and all(abs(chk.m[j] - chk.a[3, j]) < mp.mpf("1e-15") for j in range(3))    # y1 = Y4 + U4
