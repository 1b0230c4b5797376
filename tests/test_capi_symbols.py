"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/phoskin.h declares, and its pure-host
entry points (shape arithmetic, default options, argument errors) behave.  No kernel is launched here."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _header_functions():
    txt = (ROOT / "include" / "phoskin.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pk_[a-z0-9_]+)\s*\(", txt)) - {"pk_ctx", "pk_solver_opts"})


def test_header_and_binding_agree():
    from phoskintime_amd import _capi
    assert _header_functions() == sorted(_capi.SYMBOLS)


def test_library_exports_every_symbol(built_lib):
    for name in _header_functions():
        assert hasattr(built_lib, name), name
    assert built_lib.pk_version() == 200


def test_shapes(built_lib):
    assert built_lib.pk_protein_n_states(0, 4) == 6 and built_lib.pk_protein_n_params(0, 4) == 12
    assert built_lib.pk_protein_n_states(1, 14) == 16 and built_lib.pk_protein_n_params(1, 14) == 32
    assert built_lib.pk_protein_n_states(2, 4) == 17 and built_lib.pk_protein_n_params(2, 4) == 23
    assert built_lib.pk_protein_n_states(0, 30) == 32 and built_lib.pk_protein_n_params(0, 30) == 64
    assert built_lib.pk_protein_flat_len(0, 4, 14) == 79          # reference: flat(79,) for n = 4, T = 14
    assert built_lib.pk_protein_flat_len(2, 4, 14) == 79
    assert built_lib.pk_protein_flat_len(0, 4, 3) == 0 + 3 + 12   # T <= 5: the R(t5..) block is empty
    assert built_lib.pk_protein_n_states(3, 4) < 0 and built_lib.pk_protein_n_states(0, 0) < 0
    # sizes with a forward-sensitivity kernel (csrc/pk_sens.hpp)
    assert [built_lib.pk_protein_sens_available(m, n) for m, n in ((0, 1), (0, 14), (0, 15), (0, 62), (0, 63), (1, 62), (1, 63), (2, 5), (2, 7), (2, 8), (3, 1), (0, 0))] == [1, 1, 1, 1, 0, 1, 0, 1, 1, 0, 0, 0]


def test_default_opts_struct_layout(built_lib):
    from phoskintime_amd import _capi
    o = _capi.default_opts()
    assert (o.method, o.linsolve, o.rtol, o.atol, o.max_steps, o.clip_nonneg, o.normalize) == (5, 0, 1e-6, 1e-8, 100000, 1, 0)
    assert C.sizeof(_capi.SolverOpts) == 64          # 2 x i32, 4 x f64, 5 x i32 (+ 4 bytes tail padding)
    o = _capi.default_opts(method="bdf2", linsolve="dense", rtol=1e-9)
    assert (o.method, o.linsolve, o.rtol) == (1, 1, 1e-9)
    with pytest.raises(TypeError):
        _capi.default_opts(nonsense=1)


def test_null_context_is_an_error_not_a_crash(built_lib):
    assert built_lib.pk_synchronize(None) < 0
    assert built_lib.pk_solve_protein_batch(None, 0, 4, 1, None, None, 0, None, 14, None, None, None, None, 0, None, None) < 0
    assert built_lib.pk_solve_protein_sens_batch(None, 0, 4, 1, None, None, 0, None, 14, None, None, None, None, None) < 0
    assert built_lib.pk_last_error(None) == b"null context"
    built_lib.pk_destroy(None)


def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly when there is no GPU -- never route through the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from phoskintime_amd import batch
    from phoskintime_amd._capi import PhoskinError
    import numpy as np
    with pytest.raises(PhoskinError):
        batch.solve_ode_batch(0, np.ones((1, 12)), np.ones(6), 4, [0.0, 1.0])
    with pytest.raises(PhoskinError):
        batch.solve_ode_sens_batch(0, np.ones((1, 12)), np.ones(6), 4, [0.0, 1.0])
    from phoskintime_amd.models import distmod
    with pytest.raises(PhoskinError):
        distmod.solve_ode_jac(np.ones(12), np.ones(6), 4, [0.0, 1.0])
    src = "".join(p.read_text() for p in (ROOT / "phoskintime_amd").rglob("*.py"))
    assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", "")


def test_header_is_valid_plain_c(tmp_path):
    """include/phoskin.h is the C ABI: it must compile as C (no C++-isms) and expose the structs with the sizes the binding assumes."""
    import subprocess
    from phoskintime_amd import _capi
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include "phoskin.h"\nint main(void){printf("%zu %zu %zu\\n", sizeof(pk_solver_opts), sizeof(pk_network_desc), '
                   'sizeof(pk_loss_data)); return 0;}\n')
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(v) for v in out] == [C.sizeof(_capi.SolverOpts), C.sizeof(_capi.NetworkDesc), C.sizeof(_capi.LossData)]
