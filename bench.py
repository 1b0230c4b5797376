#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: ODE-solve replicas/sec, 32-state distributive model, 14 time points.

Workload (config.workload): BASELINE config 3 -- 65 536 parameter vectors theta ~ U(0, 20)^64 (the reference's config bounds,
config.toml:189-195) per GPU, models.distmod with 30 phosphosites (S = 32), y0 = 1, the reference's 14-point time grid; one
"step" = one pass of the hot path over that batch: solve (adaptive LRP12 by default, analytic Jacobian) -> clip -> trajectories [B,14,32]
written to HBM + the fused Morris scalar per replica, then -- for N > 1 -- ONE all-gather (RCCL) of the per-replica scalars.
Inputs are resident in HBM before the timed region.  Weak scaling: every rank owns its own 65 536 replicas.

Run:  python bench.py [--gpus N --steps K --warmup W]      (N > 1: under torch.distributed.run, one rank per GPU)
Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline`, `cpu_baseline` and `parity`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X FP64 vector peak (spec sheet; 256 CU x 4 SIMD x 16 DFMA lanes/clk x 2 x 2.4 GHz)


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
    ap.add_argument("--no-network", action="store_true", help="skip the secondary network-path measurement")
    ap.add_argument("--cpu-sample", type=int, default=0, help="replicas in the CPU baseline sample (0 = 128 per core)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    # a checkout without the (git-ignored) library: local rank 0 builds it (hipcc is part of the image), the others wait for the file
    lib_path = ROOT / "phoskintime_amd" / "libphoskin_hip.so"
    if not lib_path.exists():
        if local_rank == 0:
            import __graft_entry__ as graft
            graft.build()
        else:
            last = -1
            for _ in range(1800):                         # wait until the file exists and has stopped growing
                size = lib_path.stat().st_size if lib_path.exists() else -1
                if size > 0 and size == last:
                    break
                last = size
                time.sleep(2.0)
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

    model, n_sites = _capi.DIST, 30
    S, P = batch.n_states(model, n_sites), batch.n_params(model, n_sites)
    B = args.replicas
    tgrid = np.array([0.0, 0.5, 0.75, 1.0, 2.0, 4.0, 8.0, 16.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0])
    T = tgrid.size
    rng = np.random.default_rng(20260515 + 2 + 7919 * rank)          # rank 0 == the seed of tests/golden/*_c3bounds.npz
    theta_h = rng.uniform(0.0, 20.0, (B, P))
    theta = torch.as_tensor(theta_h, device=dev)
    y0 = torch.ones(S, dtype=torch.float64, device=dev)
    tt = torch.as_tensor(tgrid, device=dev)
    # two result sets, used alternately: the all-gather of step i runs on its own HIP stream while step i + 1 computes into the other set
    def new_out():
        return batch.BatchResult(sol=torch.empty((B, T, S), dtype=torch.float64, device=dev), flat=None,
                                 metric=torch.empty(B, dtype=torch.float64, device=dev),
                                 status=torch.zeros(B, dtype=torch.int32, device=dev),
                                 n_steps=torch.zeros((B, 2), dtype=torch.int32, device=dev))
    outs = [new_out(), new_out()] if use_dist else [new_out()]
    out = outs[0]
    if args.rtol is None:
        args.rtol = 1e-6 if args.method == "lrp12" else 1e-7
    if args.atol is None:
        args.atol = args.rtol * 1e-2
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

    for _ in range(args.warmup):
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
        res = {
            "metric": "ODE-solve replicas/sec (32-state distributive, 14 tp)", "value": value, "unit": "replicas/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: 65536 replicas/GPU, models.distmod n_sites=30 (S=32, P=64), theta~U(0,20), "
                                   "y0=1, 14-point grid 0..960, adaptive %s rtol=%g atol=%g, linsolve=%s; outputs sol[B,14,32] + "
                                   "Morris total_signal[B]%s" % (args.method, args.rtol, args.atol, args.linsolve,
                                                               "; 1 RCCL all-gather of Y per step" if use_dist else ""),
                       "replicas_per_gpu": B, "n_states": S, "n_params": P, "n_timepoints": T, "method": args.method,
                       "parallelism": "replica-sharded x%d" % world},
            "roofline": {"bound": "hbm", "kernel": {"lrp12": "pk::dist_fast_kernel<4, 8, 5, true, 2>", "lrp8": "pk::dist_fast_kernel<8, 4, 3, false, 1>", "rodas4": "pk::dist_fast_kernel<8, 4, 0, false, 1>"}.get(args.method, "see config") if args.linsolve == "auto" else "see config", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel_ms": kernel_ms, "algorithmic_bytes_per_replica": bytes_per_replica,
                         "note": "path is FP64 VALU-issue bound, not HBM bound (DESIGN.md); HBM fraction reported as the contract asks"},
            "solver": {"mean_accepted_steps": nst[0], "mean_rejected_steps": nst[1], "flagged_replicas": status_bad},
        }
        # HBM traffic of this very kernel + workload from the committed rocprofv3 PMC passes (tools/profile_bench.sh)
        pmc = ROOT / "profiles" / ("r01_k_final_pmc.json" if args.method == "lrp12" else "r01_g_dist_fast_lrp8_pmc.json")
        if pmc.exists() and args.method in ("lrp12", "lrp8") and args.linsolve == "auto" and B == 65536 and (args.rtol, args.atol) == ((1e-6, 1e-8) if args.method == "lrp12" else (1e-7, 1e-9)):
            pj = json.loads(pmc.read_text())
            res["roofline"]["traffic"] = pj["hbm_bytes_per_launch"]
            res["roofline"]["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH doubled)" % pmc.name
            res["roofline"]["algorithmic_bytes_per_launch"] = B * bytes_per_replica
            # the binding resource: VALU issue.  f64 VALU ops take 4 cycles per wave64 on a SIMD (16 lanes / clk)
            res["valu_issue"] = {"valu_insts_per_launch": pj["SQ_INSTS_VALU"], "lds_insts_per_launch": pj["SQ_INSTS_LDS"],
                                 "busy_frac_at_4clk_2p1GHz": pj["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.1e9 * kernel_ms * 1e-3),
                                 "note": "1024 SIMDs; clock under FP64 load ~2.1 GHz; counters from the same PMC run"}
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
        if not args.no_cpu_baseline and world == 1:
            from oracle import protein_models as pm
            cores = min(os.cpu_count() or 1, 16)
            nsamp = args.cpu_sample or 128 * cores
            rate, wall = pm.cpu_baseline(model, n_sites, theta_h[:nsamp], np.ones(S), tgrid, cores)
            res["cpu_baseline"] = {"value": rate, "unit": "replicas/s", "cores": cores, "kind": "port",
                                   "sample": "first %d replicas of the same batch; reference call shape (SciPy odeint/LSODA at default "
                                             "tolerances -> clip -> flat) on the oracle's numpy-vectorised distmod RHS, one process per core; "
                                             "wall %.1f s" % (nsamp, wall)}
        if not args.no_cpu_baseline and world == 1:
            # the SAME algorithm (LRP12 / LRP8 + arrow elimination) in scalar C on the host cores: separates what the method buys from what the GPU buys
            try:
                from oracle import lrp8_cpu
                cores = min(os.cpu_count() or 1, 16)
                nsamp2 = 2048 * cores
                rate2 = lrp8_cpu.cpu_rate(theta_h[:nsamp2], n_sites, np.ones(S), tgrid, cores, stages=(8 if args.method == "lrp8" else 12),
                                          rtol=args.rtol, atol=args.atol)
                res["cpu_same_algorithm"] = {"value": rate2, "unit": "replicas/s", "cores": cores, "kind": "port",
                                             "sample": "first %d replicas; oracle/lrp8_dist.c (gcc -O2, scalar; %s at the same tolerances), one process per core" % (nsamp2, "LRP8" if args.method == "lrp8" else "LRP12")}
            except Exception as e:
                res["cpu_same_algorithm"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1 and not args.no_network:
            # secondary evidence (NOT the metric): the network path of BASELINE configs 4 / 5 on a synthetic network of their shape
            try:
                from phoskintime_amd.global_model import NetworkEngine, synthetic
                net = synthetic.make_network(model=0)
                eng = NetworkEngine(**net)
                Xn = torch.as_tensor(synthetic.random_candidates(net, 8192, seed=1), device=dev)
                tn = np.unique(np.concatenate([net["kin_grid"], [15.0]]))
                eng.simulate_batch(Xn[:256], tn, rtol=1e-5, atol=1e-7); torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                Yn, stn, nsn = eng.simulate_batch(Xn, tn, rtol=1e-5, atol=1e-7); torch.cuda.synchronize(dev)
                dtn = time.perf_counter() - t1
                res["network_secondary"] = {"workload": "synthetic config-5-shaped network (N=100 proteins, 300 sites, S=500 states, n_var=841), "
                                                        "8192 candidates, simulate at the reference's sensitivity tolerance rtol 1e-5 / atol 1e-7",
                                            "candidates_per_s": 8192 / dtn, "ms": 1e3 * dtn, "mean_steps": float(nsn[:, 0].double().mean()),
                                            "flagged": int((stn != 0).sum()), "integrator": "ROS34PW2 Rosenbrock-W, block-diagonal Jacobian"}
                eng.close()
            except Exception as e:  # never let the secondary line break the metric
                res["network_secondary"] = {"error": repr(e)}
            # secondary evidence (NOT the metric): the other per-protein configurations BASELINE.json lists, at their sizes
            try:
                other = {}
                for label, mdl, nn, Bo in (("config1_size_distmod_n4_B65536", "distmod", 4, 65536), ("config2_succmod_n14_B4096", "succmod", 14, 4096),
                                           ("config2_size_succmod_n14_B65536", "succmod", 14, 65536), ("randmod_n4_B65536", "randmod", 4, 65536)):
                    Po, So = batch.n_params(mdl, nn), batch.n_states(mdl, nn)
                    tho = torch.as_tensor(np.random.default_rng(20260515).uniform(0.0, 20.0, (Bo, Po)), device=dev)
                    oo = batch.solve_ode_batch(mdl, tho, np.ones(So), nn, tt, want_flat=False)
                    for _ in range(50):
                        batch.solve_ode_batch(mdl, tho, np.ones(So), nn, tt, want_flat=False, out=oo)
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    for _ in range(200):
                        batch.solve_ode_batch(mdl, tho, np.ones(So), nn, tt, want_flat=False, out=oo)
                    torch.cuda.synchronize(dev)
                    dto = (time.perf_counter() - t1) / 200
                    other[label] = {"replicas_per_s": Bo / dto, "ms": 1e3 * dto, "flagged": int((oo.status != 0).sum())}
                res["other_protein_configs"] = other
            except Exception as e:
                res["other_protein_configs"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
