#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: ODE-solve replicas/sec, 32-state distributive model, 14 time points.

Workload (config.workload): BASELINE config 3 -- 65 536 parameter vectors theta ~ U(0, 20)^64 (the reference's config bounds,
config.toml:189-195) per GPU, models.distmod with 30 phosphosites (S = 32), y0 = 1, the reference's 14-point time grid; one
"step" = one pass of the hot path over that batch: solve (adaptive LRP12 by default, analytic Jacobian) -> clip -> trajectories [B,14,32]
written to HBM + the fused Morris scalar per replica, then -- for N > 1 -- ONE all-gather (RCCL) of the per-replica scalars.
Inputs are resident in HBM before the timed region.  Weak scaling: every rank owns its own 65 536 replicas.

Run:  python bench.py [--gpus N --steps K --warmup W]      (N > 1: under torch.distributed.run, one rank per GPU)
Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline`, `cpu_baseline` and `parity`.

Order of work on rank 0 at N = 1: (1) the CPU legs (cpu_baseline, cpu_same_algorithm, the single-call CPU latency) run FIRST, before this
process makes its first GPU call -- they fork worker pools, and HIP does not support fork() after initialisation; (2) the timed region;
(3) secondary, untimed-for-the-metric legs: peaks measured on this box, BASELINE config 1 as the reference runs it (one theta per call),
BASELINE config 5's network path at the optimiser's tolerance, the other per-protein sizes.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (~6.3 TB/s achievable)
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X FP64 vector peak (spec sheet; 256 CU x 4 SIMD x 16 DFMA lanes/clk x 2 x 2.4 GHz)
TGRID = np.array([0.0, 0.5, 0.75, 1.0, 2.0, 4.0, 8.0, 16.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])


def algorithmic_flops_per_step(n: int) -> int:
    """FP64 flops one LRP12 step of the arrow (distributive) system NEEDS, counted from the algorithm, S = n + 2:
    rhs 3n + 6, scaling S, factorisation 6n + 10, twelve solves 12 (4n + 6), accumulation of y_new and of the error estimate 46 S,
    error norm 6 S.  Redundant work of a layout (shadowed rows, idle padding rows) is NOT counted: this is the numerator of a roofline."""
    S = n + 2
    return (3 * n + 6) + S + (6 * n + 10) + 12 * (4 * n + 6) + 46 * S + 6 * S


def algorithmic_flops_per_step_model(model: str, n: int) -> float:
    """FP64 flops ONE LRP12 step of ONE replica needs, counted from the algorithm (never from a layout's instruction stream), per model:
      distmod   arrow elimination: `algorithmic_flops_per_step`
      succmod   Thomas on the (n + 2)-row chain: rhs 5n + 6, scaling S, factorisation 5n + 8, twelve solves 12 (5n + 6), accumulation 46 S, norm 6 S
      randmod   2^n cube states (NM) + mRNA: rhs (2n + 2) NM; twelve solves + the inverse they use --
                n <= 5: dense NM x NM inverse (Gauss-Jordan 2 NM^3) and 2 NM^2 per solve;
                n = 6, 7, 8: parity elimination (csrc/pk_rand_parity.hpp): NE = NM / 2 even states: fill NE (3n + 4 C(n,2)), inverse 2 NE^3,
                per solve 2 NE^2 + 4 n NM for the two sparse sweeps;  accumulation 46 S, norm 6 S"""
    if model == "distmod":
        return float(algorithmic_flops_per_step(n))
    S = n + 2
    if model == "succmod":
        return float((5 * n + 6) + S + (5 * n + 8) + 12 * (5 * n + 6) + 52 * S)
    NM = 1 << n
    S = NM + 1
    base = (2 * n + 2) * NM + 52 * S
    if n <= 5:
        return float(base + 2 * NM ** 3 + 12 * 2 * NM ** 2)
    NE = NM // 2
    return float(base + NE * (3 * n + 4 * (n * (n - 1) // 2)) + 2 * NE ** 3 + 12 * (2 * NE ** 2 + 4 * n * NM))


def network_algorithmic_flops_per_step(eng, stages: int = 6, solves: int = 5) -> float:
    """FP64 flops ONE step of the additive order-4 integrator (ARK4(3)6L[2]SA: 6 stage right-hand sides, 5 block solves) needs for ONE
    candidate of a linear-topology network: per stage the TF coupling 2 nnz(TF) + per protein (site sum n_s, double
    squash ~20, block rhs 6 n_s + 10); per solve the arrow / Thomas elimination of every protein block 4 n_s + 8; the stage combinations of
    the two tableaux (15 + 15 axpys of S), the two solution updates and the error norm: 90 S.  The kinase input (one CSR product per
    BUCKET, not per stage) is left out."""
    ns = np.asarray(eng._keep[2], dtype=np.float64)
    nnz_tf = float(np.asarray(eng._keep[7]).size)
    rhs = 2.0 * nnz_tf + float(np.sum(7.0 * ns + 30.0))
    solve = float(np.sum(4.0 * ns + 8.0))
    return stages * rhs + solves * solve + 90.0 * eng.S


def roofline_fp64_entry(flops_per_launch: float, kernel_ms: float, note: str, pmc_glob: str = None, kernel_match: str = None) -> dict:
    """{"achieved", "peak", "frac", ...}: algorithmic flops / measured kernel time; executed flops from a committed PMC profile of the same
    kernel when one exists (labelled as such: counters are never re-measured inside bench.py)."""
    ach = flops_per_launch / (kernel_ms * 1e-3) / 1e12
    out = {"bound": "fp64_valu", "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TFLOPS,
           "algorithmic_flops_per_launch": flops_per_launch, "kernel_ms": kernel_ms, "how": note}
    if pmc_glob:
        prof = sorted((ROOT / "profiles").glob(pmc_glob))
        if prof:
            pj = json.loads(prof[-1].read_text())
            if kernel_match is None or kernel_match in pj.get("kernel", ""):
                t = pj.get("kernel_ns_largest_dispatch", pj.get("kernel_avg_ns_rocprof", 0.0)) * 1e-9
                if t > 0 and "f64_flops_executed_per_launch" in pj:
                    out.update({"executed_tflops_in_profile": pj["f64_flops_executed_per_launch"] / t / 1e12,
                                "executed_frac_of_spec_peak_in_profile": pj["f64_flops_executed_per_launch"] / t / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                "valu_busy_frac_in_profile": pj.get("SQ_INSTS_VALU", 0.0) * 4.0 / (1024 * 2.1e9 * t),
                                "profile": "profiles/%s (workload: %s)" % (prof[-1].name, pj.get("workload", "?"))})
    return out


def wait_for_library(lib_path: Path, local_rank: int):
    """A checkout without the (git-ignored) library: local rank 0 builds it (hipcc is part of the image) and renames it into place
    atomically; the others wait for the final name, with a bounded timeout and a clear error."""
    if lib_path.exists():
        return
    if local_rank == 0:
        import __graft_entry__ as graft
        graft.build()
        return
    deadline = time.time() + 1500.0
    while not lib_path.exists():
        if time.time() > deadline:
            raise SystemExit(f"rank {local_rank}: {lib_path} did not appear within 25 minutes (local rank 0's build failed?)")
        time.sleep(1.0)


def cpu_legs(args, theta_h, n_sites, S):
    """Runs BEFORE the first GPU call of this process (fork-based worker pools)."""
    from oracle import protein_models as pm
    out = {}
    cores = min(os.cpu_count() or 1, 16)
    nsamp = args.cpu_sample or 128 * cores
    rate, wall = pm.cpu_baseline(pm.DIST, n_sites, theta_h[:nsamp], np.ones(S), TGRID, cores)
    out["cpu_baseline"] = {"value": rate, "unit": "replicas/s", "cores": cores, "kind": "port",
                           "sample": "first %d replicas of the same batch; reference call shape (SciPy odeint/LSODA at default tolerances -> clip -> "
                                     "flat) on the oracle's numpy-vectorised distmod RHS, one process per core; wall %.1f s.  The reference "
                                     "itself runs this RHS Numba-compiled (absent from this image): expect it ~2x faster than this port" % (nsamp, wall)}
    try:
        from oracle import lrp8_cpu
        nsamp2 = 2048 * cores
        rate2 = lrp8_cpu.cpu_rate(theta_h[:nsamp2], n_sites, np.ones(S), TGRID, cores, stages=(8 if args.method == "lrp8" else 12), rtol=args.rtol, atol=args.atol)
        out["cpu_same_algorithm"] = {"value": rate2, "unit": "replicas/s", "cores": cores, "kind": "port",
                                     "sample": "first %d replicas; oracle/lrp8_dist.c (gcc -O2, scalar; %s at the same tolerances), one process per core" % (nsamp2, "LRP8" if args.method == "lrp8" else "LRP12")}
    except Exception as e:
        out["cpu_same_algorithm"] = {"error": repr(e)}
    # BASELINE config 1 as the reference runs it: ONE distmod protein (4 sites), one theta per solve_ode call, on one core
    try:
        th1 = np.random.default_rng(20260515).uniform(0.05, 2.0, 12)
        pm.solve_ode(pm.DIST, th1, np.ones(6), 4, TGRID)
        t0 = time.perf_counter()
        reps = 200
        for _ in range(reps):
            pm.solve_ode(pm.DIST, th1, np.ones(6), 4, TGRID)
        out["config1_cpu_us"] = 1e6 * (time.perf_counter() - t0) / reps
    except Exception as e:
        out["config1_cpu_us"] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--replicas", type=int, default=65536, help="replicas per GPU (BASELINE config 3: 65536)")
    ap.add_argument("--rtol", type=float, default=None, help="default: 1e-6 for lrp12 (the library default), 1e-7 for the lower-order methods")
    ap.add_argument("--atol", type=float, default=None, help="default: rtol / 100")
    ap.add_argument("--linsolve", default="auto")
    ap.add_argument("--method", default="lrp12")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-network", action="store_true", help="skip the network-path leg (BASELINE config 5)")
    ap.add_argument("--no-secondary", action="store_true", help="only the metric line (profiling runs)")
    ap.add_argument("--only-network", action="store_true", help="only the network leg, printed as its own JSON line (one workload per rocprof profile)")
    ap.add_argument("--no-ramp", action="store_true", help="skip the untimed clock-ramp launches before the warm-up steps")
    ap.add_argument("--cpu-sample", type=int, default=0, help="replicas in the CPU baseline sample (0 = 128 per core)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    wait_for_library(ROOT / "phoskintime_amd" / "libphoskin_hip.so", local_rank)
    if args.rtol is None:
        args.rtol = 1e-6 if args.method == "lrp12" else 1e-7
    if args.atol is None:
        args.atol = args.rtol * 1e-2

    n_sites, S, P, T = 30, 32, 64, TGRID.size
    B = args.replicas
    rng = np.random.default_rng(20260515 + 2 + 7919 * rank)          # rank 0 == the seed of tests/golden/*_c3bounds.npz
    theta_h = rng.uniform(0.0, 20.0, (B, P))

    # ---- (1) CPU legs: before the first GPU call of this process
    cpu = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.only_network:
        cpu = cpu_legs(args, theta_h, n_sites, S)

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    # the collective path is also taken at world size 1 when launched under torch.distributed.run with PK_FORCE_COLLECTIVE=1
    # (lets a 1-GPU box exercise exactly the code the 2/4/8-GPU runs execute)
    use_dist = world > 1 or (os.environ.get("PK_FORCE_COLLECTIVE") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=dev)

    from phoskintime_amd import batch, _capi
    if args.only_network:
        print(json.dumps({"network_config5": network_leg(dev)}), flush=True)
        return
    model = _capi.DIST
    assert (batch.n_states(model, n_sites), batch.n_params(model, n_sites)) == (S, P)
    theta = torch.as_tensor(theta_h, device=dev)
    y0 = torch.ones(S, dtype=torch.float64, device=dev)
    tt = torch.as_tensor(TGRID, device=dev)

    # two result sets, used alternately: the all-gather of step i runs on its own HIP stream while step i + 1 computes into the other set
    def new_out():
        return batch.BatchResult(sol=torch.empty((B, T, S), dtype=torch.float64, device=dev), flat=None,
                                 metric=torch.empty(B, dtype=torch.float64, device=dev),
                                 status=torch.zeros(B, dtype=torch.int32, device=dev),
                                 n_steps=torch.zeros((B, 2), dtype=torch.int32, device=dev))
    outs = [new_out(), new_out()] if use_dist else [new_out()]
    out = outs[0]
    kw = dict(want_flat=False, metric="total_signal", method=args.method, linsolve=args.linsolve, rtol=args.rtol, atol=args.atol)
    main_stream = torch.cuda.current_stream(dev)
    comm_stream = torch.cuda.Stream(device=dev) if use_dist else None
    gbufs = [torch.empty(B * world, dtype=torch.float64, device=dev) for _ in outs] if use_dist else None
    gather_done = [None, None]
    counter = [0]

    def step():
        """One pass of the hot path over this rank's batch + ONE collective (all-gather of the per-replica scalars, 8 B / replica,
        RCCL) that overlaps with the next step's kernel."""
        i = counter[0] % len(outs)
        counter[0] += 1
        if use_dist and gather_done[i] is not None:
            main_stream.wait_event(gather_done[i])           # the previous gather out of this result set must be finished
        batch.solve_ode_batch(model, theta, y0, n_sites, tt, out=outs[i], **kw)
        if not use_dist:
            return outs[i].metric
        ready = torch.cuda.Event()
        ready.record(main_stream)
        comm_stream.wait_event(ready)
        with torch.cuda.stream(comm_stream):
            dist.all_gather_into_tensor(gbufs[i], outs[i].metric)
            ev = torch.cuda.Event()
            ev.record(comm_stream)
        gather_done[i] = ev
        return gbufs[i]

    def fence():
        if use_dist:
            comm_stream.synchronize()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # clock ramp (untimed, reported as `clock_ramp_launches`): the first few dozen launches after an idle period run 15-20 % slower than
    # the steady state (0.595 ms per step over 20 steps, 0.528 over 2 000); a short-K run would otherwise time the ramp,
    # not the kernel.  Then the W warm-up steps the contract asks for, then EXACTLY K timed steps.
    ramp = 0 if args.no_ramp else max(0, 300 - args.warmup)
    for _ in range(ramp + args.warmup):
        step()
    fence()
    # HIP events on the LAUNCH stream bracket the timed region: that stream carries only the solve kernels (the collective runs on its
    # own stream), so (e1 - e0) / steps is the average kernel duration the roofline figures use
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(main_stream)
    for i in range(args.steps):
        gathered = step()
    e1.record(main_stream)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kernel_ms = e0.elapsed_time(e1) / args.steps

    status_bad = int((out.status != 0).sum().item())
    nst = out.n_steps.double().mean(dim=0).tolist()
    assert gathered.shape[0] == B * world

    if rank == 0:
        total_replicas = B * world * args.steps
        value = total_replicas / elapsed
        bytes_per_replica = 8 * (P + S + T * S)                      # read theta and y0, write sol[T,S]
        achieved = B * bytes_per_replica / (kernel_ms * 1e-3) / 1e9
        kname = {"lrp12": "pk::dist_fast_kernel<4, 8, 5, true, 2>", "lrp8": "pk::dist_fast_kernel<8, 4, 3, false, 1>",
                 "rodas4": "pk::dist_fast_kernel<8, 4, 0, false, 1>"}.get(args.method, "see config") if args.linsolve == "auto" else "see config"
        res = {
            "metric": "ODE-solve replicas/sec (32-state distributive, 14 tp)", "value": value, "unit": "replicas/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "clock_ramp_launches": ramp, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: 65536 replicas/GPU, models.distmod n_sites=30 (S=32, P=64), theta~U(0,20), "
                                   "y0=1, 14-point grid 0..960, adaptive %s rtol=%g atol=%g, linsolve=%s; outputs sol[B,14,32] + "
                                   "Morris total_signal[B]%s" % (args.method, args.rtol, args.atol, args.linsolve,
                                                               "; 1 RCCL all-gather of Y per step" if use_dist else ""),
                       "replicas_per_gpu": B, "n_states": S, "n_params": P, "n_timepoints": T, "method": args.method,
                       "parallelism": "replica-sharded x%d" % world},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel_ms": kernel_ms, "algorithmic_bytes_per_replica": bytes_per_replica,
                         "algorithmic_bytes_per_launch": B * bytes_per_replica,
                         "note": "path is FP64 VALU-issue bound, not HBM bound (DESIGN.md); HBM fraction reported as the contract asks; "
                                 "the binding roofline is `roofline_fp64`"},
            "solver": {"mean_accepted_steps": nst[0], "mean_rejected_steps": nst[1], "flagged_replicas": status_bad},
        }
        # the binding roofline: FP64 vector arithmetic.  Numerator = flops the ALGORITHM needs (counted per step, times the steps the
        # kernel reports), never the instructions a layout happens to execute
        fl_step = algorithmic_flops_per_step(n_sites)
        fl_launch = B * fl_step * (nst[0] + nst[1])
        res["roofline_fp64"] = {"bound": "fp64_valu", "achieved": fl_launch / (kernel_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": fl_launch / (kernel_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS, "algorithmic_flops_per_step": fl_step,
                                "algorithmic_flops_per_launch": fl_launch, "peak_source": "spec sheet; measured value under peaks_measured"}
        # counters of this very kernel + workload from the committed rocprofv3 PMC passes (tools/profile_bench.sh): labelled as such,
        # and only attached when the run matches the profiled configuration
        prof = sorted((ROOT / "profiles").glob("r0[2-9]_*_config3_pmc.json")) or sorted((ROOT / "profiles").glob("r01_k_final_pmc.json"))      # the latest round's
        if prof and args.method == "lrp12" and args.linsolve == "auto" and B == 65536 and (args.rtol, args.atol) == (1e-6, 1e-8):
            pj = json.loads(prof[-1].read_text())
            if pj.get("kernel") == kname:
                res["roofline"]["traffic"] = pj["hbm_bytes_per_launch"]
                res["roofline"]["traffic_source"] = "committed profile profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH doubled per the gfx950 correction); not re-measured in this run" % prof[-1].name
                if "f64_flops_executed_per_launch" in pj:
                    ex = pj["f64_flops_executed_per_launch"] / (kernel_ms * 1e-3) / 1e12
                    res["roofline_fp64"].update({"executed_flops_per_launch": pj["f64_flops_executed_per_launch"], "executed_tflops": ex,
                                                 "executed_frac_of_spec_peak": ex / FP64_VALU_PEAK_TFLOPS,
                                                 "f64_share_of_valu_insts": (pj["SQ_INSTS_VALU_FMA_F64"] + pj["SQ_INSTS_VALU_ADD_F64"] + pj["SQ_INSTS_VALU_MUL_F64"] + pj["SQ_INSTS_VALU_TRANS_F64"]) / pj["SQ_INSTS_VALU"],
                                                 "executed_source": "SQ_INSTS_VALU_{FMA,ADD,MUL}_F64 of the committed profile profiles/%s x 64 lanes (FMA = 2), kernel time from this run" % prof[-1].name})
                res["valu_issue"] = {"valu_insts_per_launch": pj["SQ_INSTS_VALU"], "lds_insts_per_launch": pj["SQ_INSTS_LDS"],
                                     "busy_frac_at_4clk_2p1GHz": pj["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.1e9 * kernel_ms * 1e-3),
                                     "note": "instruction counts from the committed profile %s, kernel time from this run; 1024 SIMDs, ~2.1 GHz under FP64 load" % prof[-1].name}
        # parity on the first 64 replicas of this very batch against the committed SciPy reference trajectories
        gfile = ROOT / "tests" / "golden" / "protein_distmod_n30_c3bounds.npz"
        if gfile.exists() and args.method in ("rodas4", "lrp8", "lrp12") and B >= 64:
            g = np.load(gfile)
            if np.array_equal(g["theta"], theta_h[:64]):
                sol64 = out.sol[:64].cpu().numpy()
                tight = np.clip(g["sol_tight"], 0, None)
                res["parity"] = {"max_band_err_vs_scipy_tight": float(np.max(np.abs(sol64 - tight) / (1e-8 + 1e-6 * np.abs(tight)))),
                                 "max_abs_dy_vs_scipy_tight": float(np.max(np.abs(sol64 - tight))),
                                 "max_abs_dy_vs_scipy_default": float(np.max(np.abs(sol64 - g["sol_default"]))),
                                 "reference_default_vs_tight_band": float(np.max(np.abs(g["sol_default"] - tight) / (1e-8 + 1e-6 * np.abs(tight)))),
                                 "n": 64, "band": "rtol 1e-6 / atol 1e-8"}
        for k in ("cpu_baseline", "cpu_same_algorithm"):
            if k in cpu:
                res[k] = cpu[k]
        if world == 1 and not args.no_secondary:
            res.update(secondary_legs(args, dev, tt, cpu))
            if res.get("peaks_measured", {}).get("fp64_fma_tflops", 0) > 0:
                res["roofline_fp64"]["frac_of_measured_peak"] = res["roofline_fp64"]["achieved"] / res["peaks_measured"]["fp64_fma_tflops"]
            if res.get("peaks_measured", {}).get("hbm_copy_gbs", 0) > 0:
                res["roofline"]["frac_of_measured_peak"] = achieved / res["peaks_measured"]["hbm_copy_gbs"]
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


def network_leg(dev):
    """BASELINE config 5's path at the reference's OPTIMISATION tolerance (rtol = atol = 1e-8, config.toml:403-404): 8 192 candidates of an
    N = 100 network through simulate + objective.  The network is the reference-built one of tests/golden/netlarge_m0.npz when present
    (candidate 0 = its parameter set 0, so the band error against the reference's LSODA@1e-12 trajectory is part of the line)."""
    from phoskintime_amd.global_model import NetworkEngine, synthetic
    from phoskintime_amd.global_model.optproblem import GlobalODEBatch
    gfile = ROOT / "tests" / "golden" / "netlarge_m0.npz"
    Bn = 8192
    rngn = np.random.default_rng(20260515 + 4)
    if gfile.exists():
        g = np.load(gfile)
        eng = NetworkEngine.from_npz(g)
        base = eng.pack_params(g["c_k"][0], g["A_i"][0], g["B_i"][0], g["C_i"][0], g["D_i"][0], g["Dp_i"][0], g["E_i"][0], g["tf_scale"][0])
        tn = g["t_eval"]; src = "reference-built network of tests/golden/netlarge_m0.npz"
    else:
        g = None
        net = synthetic.make_network(model=0)
        eng = NetworkEngine(**net)
        base = synthetic.default_candidate(net)
        tn = np.unique(np.concatenate([net["kin_grid"], [15.0]])); src = "synthetic network (fixture absent)"
    X = base[None, :] * np.exp(0.5 * rngn.standard_normal((Bn, base.size)))
    X[0] = base
    Xraw = np.log(np.expm1(np.maximum(X, 1e-12)))                       # raw decision vectors (inverse softplus), as the optimiser holds them
    Xd = torch.as_tensor(Xraw, device=dev)
    lists, ld = eng.make_index_lists(tn, tn, tn[tn >= 4.0], tn)         # every protein / site at every time of its modality
    eng.free_loss(lists)
    for k in ("obs_prot", "obs_rna", "obs_pho"):
        ld[k] = np.abs(1.0 + 0.05 * rngn.standard_normal(ld[k].size))
    sl = {"c_k": None}
    defaults = dict(zip(("c_k", "A_i", "B_i", "C_i", "D_i", "Dp_i", "E_i"), np.split(base[:-1], np.cumsum([eng.n_K, eng.N, eng.N, eng.N, eng.N, eng.total_sites])))); defaults["tf_scale"] = float(base[-1])
    prob = GlobalODEBatch(eng, sl, ld, defaults, {"protein": 1.0, "rna": 1.0, "phospho": 1.0, "prior": 0.01}, tn, rtol=1e-8, atol=1e-8)
    prob.evaluate_device(Xd[:256]); torch.cuda.synchronize(dev)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    st = torch.cuda.current_stream(dev)
    t1 = time.perf_counter()
    e0.record(st)
    Y, status, nsn = eng.simulate_batch(Xd, tn, raw=True, rtol=1e-8, atol=1e-8, max_steps=prob.max_steps * tn.size, err_norm=prob.err_norm)
    e1.record(st)
    _, F = eng.objective_batch(prob.loss, Y, loss_mode=0, x=Xd, raw=True, defaults=prob.defaults, lambdas=prob.lam, status=status)
    e2.record(st)
    torch.cuda.synchronize(dev)
    dtn = time.perf_counter() - t1
    steps = nsn.double().mean(dim=0).tolist()
    out = {"workload": "BASELINE config 5 shape: 8192 raw decision vectors (defaults x log-normal 0.5), %s: N=%d proteins, %d sites, S=%d states, "
                       "n_var=%d; simulate at rtol=atol=1e-8 (config.toml:403-404) on the %d-point grid + 3-objective loss (candidates_per_s: ONE fused launch when taken; simulate_kernel_ms / objective_kernel_ms: the two-launch path, which the roofline entry is computed on)" % (src, eng.N, eng.total_sites, eng.S, eng.n_var, tn.size),
           "candidates_per_s": Bn / dtn, "wall_ms": 1e3 * dtn, "simulate_kernel_ms": e0.elapsed_time(e1), "objective_kernel_ms": e1.elapsed_time(e2),
           "mean_accepted_steps": steps[0], "mean_rejected_steps": steps[1], "flagged": int((status != 0).sum()),
           "integrator": "ARK4(3)6L[2]SA, linearly implicit on the per-protein block Jacobian (order 4, 5 block solves per step); max-norm error control; "
                         "dense two-lanes-per-protein layout, 3 waves per SIMD (pk_network_solve_arkp.hpp)",
           "algorithmic_bytes_per_candidate": 8 * (eng.n_var + eng.S + 3), "hbm_gbs_algorithmic": Bn * 8 * (eng.n_var + eng.S + 3) / (e0.elapsed_time(e1) * 1e-3) / 1e9,
           "finite_objectives": bool(torch.isfinite(F).all())}
    fl_step = network_algorithmic_flops_per_step(eng)
    out["roofline_fp64"] = roofline_fp64_entry(Bn * fl_step * (steps[0] + steps[1]), e0.elapsed_time(e1),
                                               "network_algorithmic_flops_per_step (%.0f flop per step and candidate) x accepted + rejected steps x candidates / simulate_kernel_ms" % fl_step,
                                               "r0*_network5*_pmc.json", "net_solve_ark")
    # [r3] the path GlobalODEBatch.evaluate_device takes on this topology: simulate + objectives in ONE launch, no trajectory in HBM
    try:
        kwf = dict(raw=True, rtol=1e-8, atol=1e-8, max_steps=prob.max_steps * tn.size, err_norm=prob.err_norm, loss_mode=0, defaults=prob.defaults,
                   lambdas=prob.lam, fail_value=prob.fail_value)
        fo = eng.simulate_objective_batch(prob.loss, Xd[:256], tn, **kwf); torch.cuda.synchronize(dev)
        if fo is None:
            out["fused_simulate_objective"] = {"taken": False}
        else:
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t1 = time.perf_counter(); f0.record(st)
            _, Ff, stf, nsf, _ = eng.simulate_objective_batch(prob.loss, Xd, tn, **kwf)
            f1.record(st); torch.cuda.synchronize(dev)
            dtf = time.perf_counter() - t1
            out["fused_simulate_objective"] = {"taken": True, "kernel_ms": f0.elapsed_time(f1), "wall_ms": 1e3 * dtf, "candidates_per_s": Bn / dtf,
                                               "max_rel_diff_of_F_vs_two_launches": float(((Ff - F).abs() / F.abs().clamp_min(1e-300)).max()),
                                               "same_step_counts": bool((nsf == nsn).all()),
                                               "hbm_bytes_not_written": int(Bn) * int(tn.size) * int(eng.S) * 8}
            out["two_launch_candidates_per_s"] = out["candidates_per_s"]
            out["candidates_per_s"] = Bn / dtf
            out["wall_ms_two_launches"] = out["wall_ms"]; out["wall_ms"] = 1e3 * dtf
    except Exception as e:
        out["fused_simulate_objective"] = {"error": repr(e)}
    if g is not None:
        truth = g["Y_tight"][0]; y0c = Y[0].cpu().numpy()
        out["band_err_candidate0_vs_reference_lsoda_1e-12"] = float(np.max(np.abs(y0c - truth) / (1e-8 + 1e-6 * np.abs(truth))))
        out["reference_lsoda_1e-8_band_vs_its_1e-12"] = float(np.max(np.abs(g["Y_lsoda8"][0] - truth) / (1e-8 + 1e-6 * np.abs(truth))))
    # the round-1 integrator on the same population (order-3 Rosenbrock-W, 4 block solves per step): what the order-4 method replaced
    try:
        eng.simulate_batch(Xd[:256], tn, raw=True, rtol=1e-8, atol=1e-8, method="rosw"); torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        Yr, str_, nsr = eng.simulate_batch(Xd, tn, raw=True, rtol=1e-8, atol=1e-8, max_steps=prob.max_steps * tn.size, method="rosw")
        torch.cuda.synchronize(dev)
        dtr = time.perf_counter() - t1
        out["ros34pw2_same_population"] = {"candidates_per_s": Bn / dtr, "mean_accepted_steps": float(nsr[:, 0].double().mean()), "flagged": int((str_ != 0).sum()),
                                           "max_band_between_the_two_integrators_over_population": float(((Yr - Y).abs() / (1e-8 + 1e-6 * Y.abs())).max())}
        if g is not None:
            out["ros34pw2_same_population"]["band_err_candidate0_vs_reference_lsoda_1e-12"] = float(np.max(np.abs(Yr[0].cpu().numpy() - truth) / (1e-8 + 1e-6 * np.abs(truth))))
    except Exception as e:
        out["ros34pw2_same_population"] = {"error": repr(e)}
    prob.close(); eng.close()
    return out


def morris_leg(dev):
    """BASELINE config 4: Morris screening of the network -- in its NAMED shape (128 trajectories x 200 varied parameters = 25 728 simulations:
    the 200 entries are drawn once from the flat parameter vector, `vary=`) and in the reference's own shape (global_model/sensitivity.py:
    196-215 varies EVERY entry: 128 x (n_var + 1) simulations, a superset workload)."""
    from phoskintime_amd.global_model import NetworkEngine, synthetic
    from phoskintime_amd.global_model import sensitivity as gs
    from phoskintime_amd.global_model import config as gcfg
    gfile = ROOT / "tests" / "golden" / "netlarge_m0.npz"
    if gfile.exists():
        g = np.load(gfile); eng = NetworkEngine.from_npz(g)
        base = eng.pack_params(g["c_k"][0], g["A_i"][0], g["B_i"][0], g["C_i"][0], g["D_i"][0], g["Dp_i"][0], g["E_i"][0], g["tf_scale"][0])
    else:
        net = synthetic.make_network(model=0); eng = NetworkEngine(**net); base = synthetic.default_candidate(net)
    nK, N, sites = eng.n_K, eng.N, eng.total_sites
    cut = np.cumsum([nK, N, N, N, N, sites, N])
    fitted = dict(zip(("c_k", "A_i", "B_i", "C_i", "D_i", "Dp_i", "E_i"), np.split(base[:-1], cut[:-1]))); fitted["tf_scale"] = float(base[-1])
    tp, tr = gcfg.TIME_POINTS_PROTEIN, gcfg.TIME_POINTS_RNA
    gs.run_sensitivity_batch(eng, fitted, tp, tr, tp, trajectories=2, num_levels=40, seed=1)
    torch.cuda.synchronize(dev)
    fl_step = network_algorithmic_flops_per_step(eng)
    res = {}
    vary200 = np.sort(np.random.default_rng(20260515 + 3).choice(eng.n_var, size=200, replace=False))
    for label, vary in (("named_shape_128x200", vary200), ("every_entry_varied", None)):
        t1 = time.perf_counter()
        out = gs.run_sensitivity_batch(eng, fitted, tp, tr, tp, trajectories=128, num_levels=40, seed=3, vary=vary)
        torch.cuda.synchronize(dev)
        dt_first = time.perf_counter() - t1                 # includes the one-time growth of the allocator's pool for the [B, T, S] trajectory block
        t1 = time.perf_counter()
        out = gs.run_sensitivity_batch(eng, fitted, tp, tr, tp, trajectories=128, num_levels=40, seed=3, vary=vary)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t1
        nsim = out["Y"].size
        D = len(out["problem"]["names"])
        ms = out["mean_steps"] or [0.0, 0.0]
        res[label] = {"workload": "Morris screening of the N=%d / S=%d network: 128 trajectories x (D = %d varied parameters + 1) = %d simulations at the settings of "
                                  "simulate_and_measure -> fold-change observables -> total_signal -> elementary effects on the GPU" % (N, eng.S, D, nsim),
                      "wall_s": dt, "wall_s_first_call_at_this_size": dt_first, "simulations_per_s": nsim / dt, "flagged": int((out["status"] != 0).sum()), "finite_mu_star": bool(np.isfinite(out["Si"]["mu_star"]).all()),
                      "mean_accepted_steps": ms[0], "mean_rejected_steps": ms[1],
                      "roofline_fp64": roofline_fp64_entry(nsim * fl_step * (ms[0] + ms[1]), 1e3 * dt,
                                                           "network_algorithmic_flops_per_step x steps x simulations / WALL time of the whole driver (design, simulate, observables, effects)")}
    eng.close()
    res["workload"] = "BASELINE config 4 (Morris sensitivity scan on the global_model network, 128 trajectories x 200 params), one GPU"
    return res


def lm_leg():
    """The rows-batched bounded Levenberg-Marquardt driver (paramest.normest's fits) at two sizes, each with residuals / Jacobian columns /
    normal equations formed on the GPU and on round 1's path (every `flat` vector over PCIe, numpy algebra); "auto" picks by transfer size."""
    from phoskintime_amd import batch
    from phoskintime_amd.paramest import fit_rows_batch, multistart_candidates
    out = {}
    for label, n, rows, iters in (("multistart_48_starts_distmod_n8", 8, 48, 60), ("rows_480_distmod_n8", 8, 480, 60), ("lambda_scan_480_rows_distmod_n30", 30, 480, 12)):
        P, S = 4 + 2 * n, n + 2
        rng = np.random.default_rng(20260515 + 9)
        th_true = rng.uniform(0.2, 2.0, P)
        flat = batch.solve_ode_batch("distmod", th_true[None], np.ones(S), n, TGRID, want_sol=False).flat[0].cpu().numpy()
        target = np.abs(flat * (1 + 0.02 * rng.standard_normal(flat.size)))
        lb, ub = np.zeros(P), np.full(P, 20.0)
        P0 = multistart_candidates("BENCH", rng.uniform(lb, ub), lb, ub, n_starts=rows)
        leg = {"rows": rows, "P": P, "residuals": int(flat.size), "flat_bytes_per_jacobian": rows * P * flat.size * 8}
        modes = [("device_algebra", True, "fd", "auto"), ("host_algebra_round1", False, "fd", "host")]
        if batch.sens_available("distmod", n):
            modes.insert(0, ("forward_sensitivities", "auto", "sens", "auto"))
            modes.insert(1, ("forward_sensitivities_host_lm_round2", "auto", "sens", "host"))
        for name, dev_alg, jac, lm in modes:
            fit_rows_batch("distmod", n, TGRID, P0[:4], np.ones(S), target, bounds=(lb, ub), max_iter=2, device_algebra=dev_alg, jacobian=jac, lm_algebra=lm)
            if lm == "auto" and rows * P * P >= (1 << 19):                               # warm the batched-LU path at its real shape
                fit_rows_batch("distmod", n, TGRID, P0, np.ones(S), target, bounds=(lb, ub), max_iter=1, device_algebra=dev_alg, jacobian=jac, lm_algebra=lm)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fit = fit_rows_batch("distmod", n, TGRID, P0, np.ones(S), target, bounds=(lb, ub), max_iter=iters, device_algebra=dev_alg, jacobian=jac, lm_algebra=lm)
            torch.cuda.synchronize()
            leg[name] = {"wall_ms": 1e3 * (time.perf_counter() - t1), "iterations": fit.n_iter, "solves": fit.n_solves, "launches": fit.n_launches,
                         "best_cost": float(fit.cost.min()), "median_cost": float(np.median(fit.cost))}
        leg["auto_picks"] = ("forward_sensitivities" if batch.sens_available("distmod", n) else "device_algebra" if rows * P * flat.size * 8 > (2 << 20) else "host_algebra_round1")
        out[label] = leg
    out["workload"] = ("bounded LM fits of models.distmod (2 % noise on the target); forward_sensitivities: one launch of the sensitivity kernel per Jacobian "
                       "(n_active integrations with their P tangents); the other two: forward-difference columns (n_active * P replicas) on the throughput kernels")
    return out


def secondary_legs(args, dev, tt, cpu):
    """Evidence beside the metric (never the metric): machine peaks measured here, config 1 latency, config 5 network path, other sizes."""
    from phoskintime_amd import batch, models
    res = {}
    ctx = batch.get_context()
    try:
        res["peaks_measured"] = {"hbm_copy_gbs": ctx.lib.pk_measure_hbm_gbs(ctx.handle, 2 << 30, 10),
                                 "hbm_read_gbs": ctx.lib.pk_measure_hbm_stream_gbs(ctx.handle, 2 << 30, 10, 0), "hbm_write_gbs": ctx.lib.pk_measure_hbm_stream_gbs(ctx.handle, 2 << 30, 10, 1),
                                 "fp64_fma_tflops": ctx.lib.pk_measure_fp64_fma_tflops(ctx.handle, 1 << 16),
                                 "spec": {"hbm_gbs": HBM_PEAK_GBS, "hbm_gbs_achievable_per_guide": 6300.0, "fp64_valu_tflops": FP64_VALU_PEAK_TFLOPS},
                                 "how": "pk_measure_hbm_gbs: 10 copies of 2 GiB, one 16 KiB tile per workgroup, 16 B per lane, non-temporal loads / stores (read + written bytes / time); "
                                        "pk_measure_hbm_stream_gbs: the same tiles read-only / write-only; a copy sits below the read rate (bus turnarounds); "
                                        "pk_measure_fp64_fma_tflops: 16 independent v_fma_f64 chains per lane, 4 waves per SIMD"}
    except Exception as e:
        res["peaks_measured"] = {"error": repr(e)}
    # BASELINE config 1 as the reference runs it: one distmod protein, ONE theta per call, through the drop-in models.solve_ode
    try:
        models.set_model("distmod")
        th1 = np.random.default_rng(20260515).uniform(0.05, 2.0, 12)
        y01 = np.ones(6)
        for _ in range(50):
            models.solve_ode(th1, y01, 4, TGRID)
        lat = []
        for _ in range(500):
            t1 = time.perf_counter(); models.solve_ode(th1, y01, 4, TGRID); lat.append(time.perf_counter() - t1)
        lat = np.sort(np.array(lat)) * 1e6
        res["config1_single_call"] = {"workload": "BASELINE config 1: models.distmod, 4 sites, 14 time points, ONE theta per models.solve_ode call (host arrays in, host arrays out)",
                                      "gpu_us_median": float(lat[lat.size // 2]), "gpu_us_p90": float(lat[int(0.9 * lat.size)]), "gpu_us_min": float(lat[0]),
                                      "cpu_port_us": cpu.get("config1_cpu_us"), "workspace": ctx.workspace_stats(),
                                      "note": "one call = pack into the page-locked buffer -> 1 H2D -> kernel (1 workgroup) -> 1 D2H -> sync; no hipMalloc / hipFree"}
        # the same protein's value AND parameter Jacobian in one call (models.solve_ode_jac: the jac= callable for curve_fit); the reference's
        # curve_fit differences solve_ode 1 + P times for it
        for _ in range(20):
            models.solve_ode_jac(th1, y01, 4, TGRID)
        latj = []
        for _ in range(200):
            t1 = time.perf_counter(); models.solve_ode_jac(th1, y01, 4, TGRID); latj.append(time.perf_counter() - t1)
        latj = np.sort(np.array(latj)) * 1e6
        res["config1_single_call"]["jacobian_call_gpu_us_median"] = float(latj[latj.size // 2])
        if cpu.get("config1_cpu_us"):
            res["config1_single_call"]["jacobian_cpu_port_us_derived"] = 13 * cpu["config1_cpu_us"]
            res["config1_single_call"]["jacobian_note"] = "CPU figure = (1 + P) x cpu_port_us: the 13 solve_ode calls of scipy's 2-point differencing at P = 12"
        models.set_model("randmod")
    except Exception as e:
        res["config1_single_call"] = {"error": repr(e)}
    if not args.no_network:
        try:
            res["network_config5"] = network_leg(dev)
        except Exception as e:  # never let a secondary line break the metric
            res["network_config5"] = {"error": repr(e)}
        try:
            res["network_config4"] = morris_leg(dev)
        except Exception as e:
            res["network_config4"] = {"error": repr(e)}
    try:
        res["lm_fit"] = lm_leg()
    except Exception as e:
        res["lm_fit"] = {"error": repr(e)}
    try:
        other = {}
        T_ = TGRID.size
        for label, mdl, nn, Bo in (("config1_size_distmod_n4_B65536", "distmod", 4, 65536), ("config2_succmod_n14_B4096", "succmod", 14, 4096),
                                   ("config2_size_succmod_n14_B65536", "succmod", 14, 65536), ("randmod_n4_B65536", "randmod", 4, 65536), ("randmod_n5_B16384", "randmod", 5, 16384), ("randmod_n6_B16384_one_wave_parity_kernel", "randmod", 6, 16384),
                                   ("wide_distmod_n100_B4096", "distmod", 100, 4096), ("wide_succmod_n100_B4096", "succmod", 100, 4096), ("wide_randmod_n7_B1024", "randmod", 7, 1024), ("wide_randmod_n8_B1024", "randmod", 8, 1024),
                                   ("wide_randmod_n9_B256_ncube_kernel", "randmod", 9, 256)):
            Po, So = batch.n_params(mdl, nn), batch.n_states(mdl, nn)
            tho = torch.as_tensor(np.random.default_rng(20260515).uniform(0.0, 20.0, (Bo, Po)), device=dev)
            oo = batch.solve_ode_batch(mdl, tho, np.ones(So), nn, tt, want_flat=False)
            reps = 5 if label.startswith("wide") else 20 if nn >= 5 and mdl == "randmod" else 200
            for _ in range(max(1, reps // 4)):
                batch.solve_ode_batch(mdl, tho, np.ones(So), nn, tt, want_flat=False, out=oo)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(reps):
                batch.solve_ode_batch(mdl, tho, np.ones(So), nn, tt, want_flat=False, out=oo)
            torch.cuda.synchronize(dev)
            dto = (time.perf_counter() - t1) / reps
            nso = oo.n_steps.double().mean(dim=0).tolist()
            other[label] = {"replicas_per_s": Bo / dto, "ms": 1e3 * dto, "flagged": int((oo.status != 0).sum()), "mean_steps": nso[0],
                            "hbm_gbs_algorithmic": Bo * 8 * (Po + So + T_ * So) / dto / 1e9}
            if not (mdl == "randmod" and nn >= 9):
                fls = algorithmic_flops_per_step_model(mdl, nn)
                prof = {"wide_randmod_n8_B1024": ("r03_*rand8_pmc.json", "rand_parity"), "wide_randmod_n7_B1024": ("r03_*rand7_pmc.json", "rand_parity")}.get(label, (None, None))
                other[label]["roofline_fp64"] = roofline_fp64_entry(Bo * fls * (nso[0] + nso[1]), 1e3 * dto, "algorithmic_flops_per_step_model(%s, %d) = %.0f x steps x replicas / wall per launch" % (mdl, nn, fls), *prof)
        # forward sensitivities: flat and d flat / d theta of every replica from one launch (csrc/pk_sens.hpp)
        for label, mdl, nn, Bo in (("sensitivities_distmod_n8_B65536", "distmod", 8, 65536), ("sensitivities_randmod_n4_B4096", "randmod", 4, 4096),
                                   ("sensitivities_distmod_n30_B4096", "distmod", 30, 4096), ("sensitivities_succmod_n30_B4096", "succmod", 30, 4096)):
            Po, So = batch.n_params(mdl, nn), batch.n_states(mdl, nn)
            tho = torch.as_tensor(np.random.default_rng(20260515).uniform(0.2, 2.0, (Bo, Po)), device=dev)
            batch.solve_ode_sens_batch(mdl, tho[:64], np.ones(So), nn, tt)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(3):
                rs = batch.solve_ode_sens_batch(mdl, tho, np.ones(So), nn, tt)
            torch.cuda.synchronize(dev)
            dto = (time.perf_counter() - t1) / 3
            nss = rs.n_steps.double().mean(dim=0).tolist()
            fls = algorithmic_flops_per_step_model(mdl, nn) * (1 + Po)           # the same step for the state and for every tangent column
            prof = ("r03_*sens_rows_pmc.json", "sens_rows") if label == "sensitivities_distmod_n30_B4096" else ("r02_g_sens_pmc.json", "sens_kernel") if label == "sensitivities_distmod_n8_B65536" else (None, None)
            other[label] = {"jacobians_per_s": Bo / dto, "ms": 1e3 * dto, "columns": Po, "flagged": int((rs.status != 0).sum()), "mean_steps": nss[0],
                            "roofline_fp64": roofline_fp64_entry(Bo * fls * (nss[0] + nss[1]), 1e3 * dto, "(1 + P) x algorithmic_flops_per_step_model x steps x replicas / wall per launch", *prof)}
        res["other_protein_configs"] = other
    except Exception as e:
        res["other_protein_configs"] = {"error": repr(e)}
    return res


if __name__ == "__main__":
    main()
