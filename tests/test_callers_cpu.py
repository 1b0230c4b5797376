"""CPU: host logic of the reference-signature callers against fixtures made by running the reference itself
(tools/make_golden_normest.py: models/weights.py, paramest/identifiability/ci.py), plus the vectorised knock-out mask."""
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).resolve().parent / "golden"
PINS = sorted(GOLD.glob("pins_normest_*.npz"))


def test_pin_inventory():
    assert [f.name for f in PINS] == ["pins_normest_distmod.npz", "pins_normest_randmod.npz"]
    for f in PINS:
        np.load(f, allow_pickle=False)


def write_tables(tmp_path, g):
    """The two measurement tables of models/weights.get_protein_weights for the fixture's synthetic gene (same numbers the reference read)."""
    n, stds, gene = int(g["n"]), g["stds"], str(g["gene"])
    sites = [f"S_{10 * (i + 1)}" for i in range(n)]
    scols = [f"x{i}_std" for i in range(1, 15)]
    rows1 = [",".join(["GeneID", "Psite"] + scols), ",".join([gene, ""] + [repr(float(v)) for v in stds[0]])]
    rows1 += [",".join([gene, s] + [repr(float(v)) for v in stds[1 + i]]) for i, s in enumerate(sites)]
    rows1.append(",".join(["OTHER", "T_5"] + ["0.5"] * 14))
    rows2 = [",".join(["GeneID", "Psite"] + [f"x{i}" for i in range(1, 15)])] + [",".join([gene, s] + ["1.0"] * 14) for s in sites]
    rows2.append(",".join(["OTHER", "T_5"] + ["1.0"] * 14))
    p1, p2 = tmp_path / "input1_wstd.csv", tmp_path / "input2.csv"
    p1.write_text("\n".join(rows1) + "\n"); p2.write_text("\n".join(rows2) + "\n")
    return p1, p2


@pytest.mark.parametrize("f", PINS, ids=lambda f: f.stem)
def test_weights_match_the_reference(f, tmp_path, monkeypatch):
    from phoskintime_amd import config
    from phoskintime_amd.models import weights as w
    g = np.load(f)
    n, t = int(g["n"]), g["t"]
    np.testing.assert_allclose(w.early_emphasis(g["pr_data"], g["p_data"], t, n), g["early_emphasis"], rtol=1e-15, atol=0)
    np.testing.assert_allclose(w.early_emphasis(g["pr_data"][0], g["p_data"], t, n), g["early_emphasis"], rtol=1e-15, atol=0)      # 1-D protein row
    p1, p2 = write_tables(tmp_path, g)
    np.testing.assert_array_equal(w.get_protein_weights(str(g["gene"]), p1, p2), g["protein_weights"])
    monkeypatch.setattr(config, "INPUT1_WSTD_PATH", str(p1)); monkeypatch.setattr(config, "INPUT2_PATH", str(p2))
    np.testing.assert_array_equal(w.get_protein_weights(str(g["gene"])), g["protein_weights"])                                       # configured default paths
    with pytest.raises(ValueError):
        w.get_protein_weights("NOSUCHGENE", p1, p2)
    np.testing.assert_array_equal(w.full_weight(np.arange(3.0), True, 2), g["fw"])
    P = g["p0"].size
    for reg in (1, 0):
        monkeypatch.setattr(config, "USE_CUSTOM_WEIGHTS", True)
        opts = w.get_weight_options(g["target"], t, n, bool(reg), P, g["early_emphasis"], g["protein_weights"])
        assert list(opts) == [str(k) for k in g[f"wo_keys_reg{reg}"]]
        lens = g[f"wo_lens_reg{reg}"]
        off = np.concatenate([[0], np.cumsum(lens)])
        for i, k in enumerate(opts):
            np.testing.assert_allclose(opts[k], g[f"wo_vals_reg{reg}"][off[i]:off[i + 1]], rtol=2e-15, atol=0, err_msg=k)
        monkeypatch.setattr(config, "USE_CUSTOM_WEIGHTS", False)
        assert list(w.get_weight_options(g["target"], t, n, bool(reg), P, g["early_emphasis"], g["protein_weights"])) == [str(k) for k in g[f"wo_default_keys_reg{reg}"]]


@pytest.mark.parametrize("f", PINS, ids=lambda f: f.stem)
def test_confidence_intervals_match_the_reference(f):
    from phoskintime_amd.paramest.identifiability import confidence_intervals
    g = np.load(f)
    r = confidence_intervals("G", g["ci_popt"], g["ci_pcov"], g["ci_target"], g["ci_model"], alpha_val=float(g["ci_alpha"]))
    for k in ("beta_hat", "se_lin", "df_lin", "t_stat", "pval", "qt_lin", "lwr_ci", "upr_ci"):
        np.testing.assert_allclose(np.asarray(r[k], float), g[f"ci_{k}"], rtol=1e-13, atol=0, err_msg=k)
    assert confidence_intervals("G", g["ci_popt"], None, g["ci_target"], g["ci_model"]) is None


def _apply_knockout_loop(base, targets, n):
    """Statement-level semantics of the reference helper (knockout/helper.py:5-36), as a test oracle for the mask version."""
    p = np.array(base, float)
    if targets.get('transcription', False):
        p[0] = 0.0
    if targets.get('translation', False):
        p[2] = 0.0
    k = targets.get('phosphorylation', None)
    if k is True:
        p[4:4 + n] = 0.0
    elif isinstance(k, (list, tuple)):
        for i in k:
            if 0 <= i < n:
                p[4 + i] = 0.0
    return p


def test_knockout_mask_equals_the_statementwise_rule():
    from phoskintime_amd.knockout import apply_knockout, generate_knockout_combinations, _apply_knockout
    assert apply_knockout is _apply_knockout
    rng = np.random.default_rng(0)
    for n in (1, 3, 6):
        base = rng.uniform(0.5, 2.0, 4 + 2 * n)
        combos = generate_knockout_combinations(n)
        assert len(combos) == 4 * (n + 2)
        extra = [{"phosphorylation": [0, n + 5, -1]}, {"phosphorylation": (n - 1,)}, {}, {"transcription": True, "phosphorylation": []}, {"phosphorylation": False}]
        for c in combos + extra:
            got = apply_knockout(base, c, n)
            np.testing.assert_array_equal(got, _apply_knockout_loop(base, c, n))
            assert got is not base


def test_caller_names_are_importable_without_a_gpu():
    """The reference's import lines, with the package prefix."""
    from phoskintime_amd.paramest.normest import normest, find_best_lambda, worker_find_lambda, _curve_fit_multistart      # noqa: F401
    from phoskintime_amd.paramest.toggle import estimate_parameters                                                          # noqa: F401
    from phoskintime_amd.paramest.core import process_gene, process_gene_wrapper                                             # noqa: F401
    from phoskintime_amd.sensitivity import sensitivity_analysis                                                             # noqa: F401
    from phoskintime_amd.sensitivity.analysis import _perturb_solve, _sensitivity_analysis                                   # noqa: F401
    from phoskintime_amd.global_model.sensitivity import run_sensitivity_analysis, _worker_simulation                        # noqa: F401
    from phoskintime_amd.global_model.lossfn import LOSS_FN, loss_function_noncomb, loss_function_comb                       # noqa: F401
    import inspect
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(normest) == ["gene", "pr_data", "p_data", "r_data", "init_cond", "num_psites", "time_points", "bounds", "bootstraps", "use_regularization"]
    assert sig(_sensitivity_analysis)[:10] == ["pr_data", "p_data", "rna_data", "popt", "time_points", "num_psites", "psite_labels", "state_labels", "init_cond", "gene"]
    assert sig(run_sensitivity_analysis)[:5] == ["sys", "idx", "fitted_params", "output_dir", "metric"]
    assert sig(LOSS_FN) == ["Y", "p_prot", "t_prot", "obs_prot", "w_prot", "p_rna", "t_rna", "obs_rna", "w_rna", "p_pho", "s_pho", "t_pho", "obs_pho", "w_pho",
                            "prot_map", "prot_base_idx", "rna_base_idx", "pho_base_idx"]
    assert sig(find_best_lambda)[:9] == ["gene", "target", "p0", "time_points", "free_bounds", "init_cond", "num_psites", "p_data", "pr_data"]
    assert sig(_curve_fit_multistart) == ["gene", "model_func", "time_points", "target_fit", "base_p0", "free_bounds", "sigma", "init_cond", "num_psites", "target",
                                          "n_starts", "jitter_frac", "maxfev", "seed"]


def test_pointwise_losses_host_versions():
    from phoskintime_amd.global_model import lossfn as lf
    d = np.array([-30.0, -0.7, -0.2, 0.0, 0.3, 0.5, 2.0, 25.0])
    np.testing.assert_allclose(lf.huber(d, 0.5), np.where(np.abs(d) <= 0.5, 0.5 * d * d, 0.5 * (np.abs(d) - 0.25)))
    np.testing.assert_allclose(lf.pseudo_huber(d, 0.5), 0.25 * (np.sqrt(1 + (d / 0.5) ** 2) - 1))
    np.testing.assert_allclose(lf.log_cosh(d)[1:-1], np.log(np.cosh(d[1:-1])))
    assert lf.log_cosh(np.array([25.0]))[0] == 25.0 - 0.69314718056
    np.testing.assert_allclose(lf.cauchy_loss(d, 1.0), np.log1p(d * d))
    np.testing.assert_allclose(lf.geman_mcclure(d, 1.0), d * d / (d * d + 1))
    np.testing.assert_allclose(lf.charbonnier(d, 1e-3), np.sqrt(d * d + 1e-6) - 1e-3)
    np.testing.assert_allclose(lf.poisson_scaled_mse(d, 2.0, 1e-6), d * d / (2.0 + 1e-6))
    np.testing.assert_array_equal(lf.sq(d), d * d)
