#!/usr/bin/env python3
"""Golden vectors for the reference-signature callers, made by RUNNING THE REFERENCE ITSELF (build container only).

    python tools/make_golden_normest.py randmod      -> tests/golden/pins_normest_randmod.npz
    python tools/make_golden_normest.py distmod      -> tests/golden/pins_normest_distmod.npz
    python tools/make_golden_normest.py all

Same recipe as tools/make_golden.py / make_golden_pins.py (writable temp copy of the tree, identity ``numba``, ``tomllib`` -> ``tomli``,
empty stand-ins for the plotting-only packages).  Two things the verbatim ``paramest.normest.normest`` call needs and the tree lacks are
DATA, written into the temp copy (never the repo, never /root/reference):

  * ``processing/input1_wstd.csv`` and ``data/input2.csv`` -- the per-gene measurement uncertainties ``models.weights.get_protein_weights``
    reads (models/weights.py:79-145).  A synthetic gene "GENEX" with a protein row and ``n`` phospho rows is written; the same numbers
    are stored in the fixture so that the drop-in reads the same weights;
  * ``[ode] model = ...`` in the temp copy's config.toml selects the model under test (the reference fixes ODE_MODEL at import).

``plotting.Plotter`` (seaborn bar chart of the confidence intervals, normest.py:545-548) is replaced by a no-op in the parent process: it
is not on the numerical path.  ``find_best_lambda`` / ``_curve_fit_multistart`` are wrapped by recorders that pass every argument
through and store what came back.  Nothing of the reference is copied: only numbers and name strings are written.

Pinned:
  models/weights.py:10-76, 79-145, 148-240      early_emphasis, get_protein_weights, full_weight, get_weight_options (17 options + the
                                                 USE_CUSTOM_WEIGHTS = False selection)
  paramest/identifiability/ci.py:10-84          confidence_intervals
  paramest/normest.py:22-115, 118-165, 328-563  worker_find_lambda scores per lambda, find_best_lambda's pick, normest's return tuple
  sensitivity/analysis.py:178-195               _perturb_solve
"""
import sys, pathlib, importlib, subprocess, re
import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
OUT = HERE.parent / "tests" / "golden"
GENE = "GENEX"


def write_inputs(tree: pathlib.Path, n: int, stds: np.ndarray):
    """input1_wstd.csv / input2.csv for one synthetic gene: protein row (Psite empty) + n phospho rows, x1_std..x14_std = stds."""
    (tree / "processing").mkdir(exist_ok=True)
    (tree / "data").mkdir(exist_ok=True)
    sites = [f"S_{10 * (i + 1)}" for i in range(n)]
    cols = [f"x{i}" for i in range(1, 15)]
    scols = [f"x{i}_std" for i in range(1, 15)]
    with open(tree / "processing" / "input1_wstd.csv", "w") as f:
        f.write(",".join(["GeneID", "Psite"] + scols) + "\n")
        f.write(",".join([GENE, ""] + [repr(float(v)) for v in stds[0]]) + "\n")
        for i, s in enumerate(sites):
            f.write(",".join([GENE, s] + [repr(float(v)) for v in stds[1 + i]]) + "\n")
        f.write(",".join(["OTHER", "T_5"] + ["0.5"] * 14) + "\n")
    with open(tree / "data" / "input2.csv", "w") as f:
        f.write(",".join(["GeneID", "Psite"] + cols) + "\n")
        for i, s in enumerate(sites):
            f.write(",".join([GENE, s] + ["1.0"] * 14) + "\n")
        f.write(",".join(["OTHER", "T_5"] + ["1.0"] * 14) + "\n")
    return sites


def main(model: str):
    import make_golden as mg
    import make_golden_pins as mp
    # the reference fixes ODE_MODEL at import from config.toml: select it in the temp copy before anything is imported
    def select_model(tree):
        cfgf = pathlib.Path(tree) / "config.toml"
        txt, k = re.subn(r'(?m)^model = "randmod"$', f'model = "{model}"', cfgf.read_text(), count=1)
        assert k == 1
        cfgf.write_text(txt)
    mods, cfg, tmp = mg.import_reference(prepare=select_model)
    mp.add_stubs()
    tree = tmp / "ref"
    n = 2
    rng = np.random.default_rng(515 + len(model))
    stds = rng.uniform(0.05, 0.4, (1 + n, 14))
    write_inputs(tree, n, stds)

    ne = importlib.import_module("paramest.normest")
    we = importlib.import_module("models.weights")
    ci = importlib.import_module("paramest.identifiability.ci")
    sa = importlib.import_module("sensitivity.analysis")
    assert ne.ODE_MODEL == model, ne.ODE_MODEL
    rmod = mods[model]
    tp = mg.TIME_POINTS
    d = dict(model=np.array(model), gene=np.array(GENE), n=n, stds=stds, t=tp)

    # --- synthetic data of one gene: a true parameter vector, 3 % multiplicative noise
    P = 4 + n + ((1 << n) - 1 if model == "randmod" else n)
    S = 2 + ((1 << n) - 1 if model == "randmod" else n)
    th_true = rng.uniform(0.2, 1.5, P)
    y0 = np.ones(S)
    _, flat_true = rmod.solve_ode(th_true, y0, n, tp)
    flat_noisy = np.abs(flat_true * (1 + 0.03 * rng.standard_normal(flat_true.size)))
    r_data = flat_noisy[:9].reshape(1, 9)
    pr_data = flat_noisy[9:23].reshape(1, 14)
    p_data = flat_noisy[23:].reshape(n, 14)
    bounds = {"A": (0.0, 20.0), "B": (0.0, 20.0), "C": (0.0, 20.0), "D": (0.0, 20.0), "S(i)": (0.0, 20.0), "D(i)": (0.0, 20.0)}
    d.update(theta_true=th_true, y0=y0, r_data=r_data, pr_data=pr_data, p_data=p_data,
             bounds_keys=np.array(list(bounds)), bounds_vals=np.array([bounds[k] for k in bounds]))

    # --- models/weights.py
    ew = we.early_emphasis(pr_data, p_data, tp, n)
    gw = we.get_protein_weights(GENE)
    target = np.concatenate([r_data.flatten(), pr_data.flatten(), p_data.flatten()])
    d.update(early_emphasis=ew, protein_weights=gw, target=target)
    for reg in (True, False):
        we.USE_CUSTOM_WEIGHTS = True
        opts = we.get_weight_options(target, tp, n, reg, P, ew, gw)
        d[f"wo_keys_reg{int(reg)}"] = np.array(list(opts))
        # ragged: the time-index schemes are 14 n long where the data block is 14 (1 + n) (models/weights.py:182, 204-206) -- kept as is
        d[f"wo_lens_reg{int(reg)}"] = np.array([len(opts[k]) for k in opts])
        d[f"wo_vals_reg{int(reg)}"] = np.concatenate([np.asarray(opts[k], float) for k in opts])
        we.USE_CUSTOM_WEIGHTS = False
        opts = we.get_weight_options(target, tp, n, reg, P, ew, gw)
        d[f"wo_default_keys_reg{int(reg)}"] = np.array(list(opts))
    d["fw"] = we.full_weight(np.arange(3.0), True, 2)

    # --- paramest/identifiability/ci.py
    rg = np.random.default_rng(9)
    A_ = rg.standard_normal((P, P)); pcov_t = A_ @ A_.T / P + 0.1 * np.eye(P)
    popt_t = rg.uniform(0.1, 2.0, P); tgt_t = rg.uniform(0.5, 2.0, 40); mdl_t = tgt_t + 0.05 * rg.standard_normal(40)
    res = ci.confidence_intervals(GENE, popt_t, pcov_t, tgt_t, mdl_t, alpha_val=0.95)
    d.update(ci_popt=popt_t, ci_pcov=pcov_t, ci_target=tgt_t, ci_model=mdl_t, ci_alpha=0.95,
             **{f"ci_{k}": np.asarray(res[k], float) for k in ("beta_hat", "se_lin", "df_lin", "t_stat", "pval", "qt_lin", "lwr_ci", "upr_ci")})
    assert ci.confidence_intervals(GENE, popt_t, None, tgt_t, mdl_t) is None

    # --- sensitivity/analysis.py:178-195 _perturb_solve
    X = tuple(float(v) for v in th_true * 1.1)
    i, sol, flat, Yv = sa._perturb_solve((7, X, y0, n, tp))
    d.update(ps_X=np.array(X), ps_i=i, ps_sol=sol, ps_flat=flat, ps_Y=Yv, ps_metric=np.array(sa.Y_METRIC))

    # --- paramest/normest.py: the verbatim call, with recorders
    rec = {}
    real_fbl, real_ms = ne.find_best_lambda, ne._curve_fit_multistart

    def fbl(*a, **k):
        out = real_fbl(*a, **k)
        rec["fbl_p0"] = np.array(a[2], float); rec["fbl_out"] = out
        return out

    def ms(*a, **k):
        out = real_ms(*a, **k)
        rec["ms_sigma"] = np.asarray(k["sigma"], float); rec["ms_target_fit"] = np.asarray(k["target_fit"], float)
        rec["ms_popt"], rec["ms_pcov"], rec["ms_score"] = np.asarray(out[0], float), out[1], float(out[2])
        return out

    class _NoPlot:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None
    ne.find_best_lambda, ne._curve_fit_multistart, ne.Plotter = fbl, ms, _NoPlot
    # per-lambda scores of the scan (worker_find_lambda, in-process, verbatim) for the drop-in's scan to be compared against
    lb_full = [0.0] * P; ub_full = [20.0] * P
    if model == "randmod":
        fb = ([np.log(max(b, 1e-8)) for b in lb_full], [np.log(b) for b in ub_full])
    else:
        fb = (lb_full, ub_full)
    np.random.seed(42)
    p0 = np.array([np.random.uniform(low=l, high=u) for l, u in zip(*fb)])
    lams = np.logspace(-2, 0, 10)
    scan = [ne.worker_find_lambda(float(l), GENE, target, p0, tp, fb, y0, n, p_data, pr_data) for l in lams[[0, 4, 9]]]
    d.update(scan_lambdas=lams[[0, 4, 9]], scan_scores=np.array([s[1] for s in scan]), scan_keys=np.array([s[2] for s in scan]), p0=p0)
    est, fits, errs, regterm = ne.normest(GENE, pr_data, p_data, r_data, y0, n, tp, bounds, 0)
    np.testing.assert_array_equal(rec["fbl_p0"], p0)
    d.update(lambda_reg=float(rec["fbl_out"][0]), lambda_weight=np.array(rec["fbl_out"][1]), ms_sigma=rec["ms_sigma"], ms_target_fit=rec["ms_target_fit"],
             ms_popt=rec["ms_popt"], ms_pcov=(np.zeros((0, 0)) if rec["ms_pcov"] is None else rec["ms_pcov"]), ms_score=rec["ms_score"],
             est_params=np.stack(est), fit_sol=fits[0][0], fit_flat=fits[0][1], error_vals=np.array(errs), regularization_term=float(regterm))
    # the confidence-interval file normest wrote (normest.py:535-543)
    import pandas as pd
    cif = pd.read_csv(pathlib.Path(ne.OUT_DIR) / f"{GENE}_confidence_intervals.csv")
    d.update(ci_file_columns=np.array(list(cif.columns)), ci_file_params=np.array(list(cif["Parameter"])), ci_file_estimate=cif["Estimate"].values.astype(float))
    # with bootstraps (global NumPy state seeded by normest itself, normest.py:386): the mean of 3 refits replaces the estimate
    est_b, fits_b, errs_b, reg_b = ne.normest(GENE, pr_data, p_data, r_data, y0, n, tp, bounds, 3)
    d.update(boot_est_params=np.stack(est_b), boot_error_vals=np.array(errs_b), boot_regularization_term=float(reg_b))
    np.savez_compressed(OUT / f"pins_normest_{model}.npz", **d)
    print(f"wrote pins_normest_{model}.npz: lambda {d['lambda_reg']}, weight {rec['fbl_out'][1]}, multistart score {rec['ms_score']:.6g}, "
          f"error {errs[0]:.6g}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "all":
        procs = [subprocess.Popen([sys.executable, __file__, m]) for m in ("randmod", "distmod")]
        sys.exit(max(p.wait() for p in procs))
    main(which)
