#!/usr/bin/env python3
"""Benchmark of the numeric solve path on MI355X (contract: see the task statement / DESIGN.md "Measurement").

Workload (BASELINE.json configs[1], "C2"): 2-D 5-point Laplacian 1024 x 1024 (n = 1 048 576), AMD ordering
(order = 0), Cholesky factor L computed by the GPU numeric Cholesky.  One STEP = lsolve + ltsolve of a block of
`--nrhs` right-hand sides per GPU (row-major, resident in HBM), i.e. the solve phase of cholsol applied to a
RHS block.  value = algorithmic GB/s summed over all ranks:
    bytes_step = 2 * (nnz(L) * 12 + (n + 1) * 4 + 16 * n * nrhs)          
Multi-GPU (one process per GPU, torch.distributed/RCCL): rank 0 factors, the factor is broadcast once
(untimed set-up), every rank solves its own RHS block -> weak scaling, no collective in the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def bytes_trisolve(n, nnz, k):
    return nnz * 12 + (n + 1) * 4 + 16 * n * k


def measured_traffic(cfg):
    """HBM bytes of one step from the committed rocprofv3 --pmc passes (profiles/*_summary.json, made by
    tools/pmc.sh + tools/collect_profiles.py on this same command), or None when no pass matches this configuration
    (same matrix, right-hand sides and programme): counters cannot be read from inside the timed process."""
    import glob
    keys = ("n", "nnz_L", "nrhs_per_gpu", "lsolve_tasks", "lsolve_launches", "ltsolve_launches")
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        t, c = d.get("traffic"), d.get("config", {})
        if t and all(c.get(k) == cfg.get(k) for k in keys):
            best = t["hbm_bytes_per_step"]
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nrhs", type=int, default=256)
    ap.add_argument("--grid", type=int, default=1024, help="Laplacian grid side (1024 = the headline configuration)")
    ap.add_argument("--cpu-rhs", type=int, default=160, help="right-hand sides of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--skip-extras", action="store_true", help="skip the single-RHS cholsol timing")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # rehearsal hook for a one-GPU box: RSP_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo, so that the whole
    # multi-rank flow (broadcast of the factor, barriers, max over ranks) can be exercised without a second GPU
    rehearsal = os.environ.get("RSP_BENCH_REHEARSAL") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import rsparse_amd as rs
    from rsparse_amd import synth
    rs.lib().rsp_set_device(dev.index)

    n = args.grid * args.grid
    nrhs = args.nrhs
    setup = {}

    # ---- set-up (untimed): rank 0 orders, analyses and factors on its GPU ------------------------------------
    extras = {}
    if rank == 0:
        t0 = time.time()
        A = synth.laplacian_2d(args.grid)
        setup["generate_s"] = time.time() - t0
        t0 = time.time()
        S = rs.schol(A, 0)
        setup["schol_host_s"] = time.time() - t0
        t0 = time.time()
        F = rs.CholHandle(A, S)
        setup["chol_analyze_plus_factor_s"] = time.time() - t0
        info = F.info()
        L = F.factor()
        # single right-hand side cholsol solve phase (P, lsolve, ltsolve, P'), factor resident
        xstar = 2.0 * synth.splitmix64(1, n) - 1.0
        import scipy.sparse as sp
        Asp = sp.csc_matrix((A[4], A[3], A[2]), shape=(n, n))
        b = Asp @ xstar
        x1 = b.copy()
        F.solve(x1)
        extras["residual_1rhs"] = float(np.linalg.norm(Asp @ x1 - b) / np.linalg.norm(b))
        extras["x_error_1rhs"] = float(np.max(np.abs(x1 - xstar)) / np.max(np.abs(xstar)))
        if not args.skip_extras:
            xb = torch.tensor(b, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            F.solve_dev(xb.data_ptr(), rs.api.COL_MAJOR, n, 1, True, st)
            torch.cuda.synchronize()
            reps = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                F.solve_dev(xb.data_ptr(), rs.api.COL_MAJOR, n, 1, True, st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            extras["cholsol_solve_1rhs_ms"] = ms
            extras["cholsol_solves_per_s_1rhs"] = 1000.0 / ms
            extras["trisolve_1rhs_GBps"] = 2 * bytes_trisolve(n, info["lnz"], 1) / (ms * 1e-3) / 1e9
            del xb
            # the literal configs[1]: ONE cholsol call from scratch (ordering, analysis, factorisation, solve, PCIe)
            xe = b.copy()
            t0 = time.time()
            rs.cholsol(A, xe, 0)
            extras["cholsol_one_shot_s"] = time.time() - t0
            extras["cholsol_one_shot_x_error"] = float(np.max(np.abs(xe - xstar)) / np.max(np.abs(xstar)))
        fi = F.info()
        extras["chol_factor_ms"] = fi["factor_ms"]
        extras["chol_flops"] = fi["flops"]
        extras["chol_bytes"] = fi["bytes_chol"]
        extras["chol_GFLOPs"] = fi["flops"] / (fi["factor_ms"] * 1e-3) / 1e9
        extras["chol_GBps"] = fi["bytes_chol"] / (fi["factor_ms"] * 1e-3) / 1e9
        extras["chol_levels"] = fi["chol_levels"]
        extras["supernodes"] = fi["supernodes"]
        del F
        Lp, Li, Lx = L.p, L.i.astype(np.int64), L.x

    # ---- the factor travels to the other GPUs once (RCCL broadcast over xGMI), untimed -------------------------
    if world > 1:
        from rsparse_amd import sharding
        t0 = time.time()
        Lall = sharding.broadcast_factor((n, n, Lp, Li, Lx) if rank == 0 else None, 0, dev, dist)
        _, _, Lp, Li, Lx = Lall
        torch.cuda.synchronize()
        setup["broadcast_s"] = time.time() - t0
    lnz = len(Lx)

    t0 = time.time()
    Lt = (n, n, Lp, Li, Lx)
    lower = rs.TriHandle(Lt, rs.api.LSOLVE)
    upper = rs.TriHandle(Lt, rs.api.LTSOLVE)
    setup["tri_analyze_s"] = time.time() - t0

    # ---- RHS block of this rank, resident in HBM ---------------------------------------------------------------
    g = torch.Generator(device=dev)
    g.manual_seed(2 + rank)
    X = torch.rand((n, nrhs), dtype=torch.float64, device=dev, generator=g) * 2.0 - 1.0
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        # lsolve then ltsolve on the same block; the library pipelines column parts of the block over several streams
        rs.solve_pair_dev(lower, upper, X.data_ptr(), nrhs, nrhs, stream)

    # correctness of the timed path itself (rank 0, first column): L L' y = b against scipy's triangular solves
    check = {}
    if rank == 0:
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla
        b0 = X[:, 0].cpu().numpy().copy()
        step()
        torch.cuda.synchronize()
        y = X[:, 0].cpu().numpy()
        Lsp = sp.csc_matrix((Lx, Li, Lp), shape=(n, n))
        r = Lsp @ (Lsp.T @ y) - b0
        check["LLt_residual_col0"] = float(np.linalg.norm(r) / np.linalg.norm(b0))
    else:
        step()
    for _ in range(max(0, args.warmup - 1)):
        step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t_start
    ev_ms = e0.elapsed_time(e1)
    if world > 1:
        tt = torch.tensor([t_local], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())
        dist.barrier()
    else:
        t_max = t_local

    if rank == 0:
        bytes_step = 2 * bytes_trisolve(n, lnz, nrhs)
        ms_per_step = t_max * 1e3 / args.steps
        value = world * bytes_step / (t_max / args.steps) / 1e9
        achieved = bytes_step / (ev_ms * 1e-3 / args.steps) / 1e9  # HIP events on the launch stream, this rank
        li, ui = lower.info(), upper.info()
        out = {
            "metric": "triangular-solve GB/s",
            "value": value,
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"C2: 2D 5-point Laplacian {args.grid}x{args.grid}, cholsol(order=0) solve phase "
                            f"(lsolve+ltsolve) on {nrhs} RHS per GPU, factor from the GPU numeric Cholesky",
                "n": n, "nnz_L": lnz, "nrhs_per_gpu": nrhs, "rhs_layout": "row-major",
                "bytes_per_step": bytes_step,
                "lsolve_launches": li["launches"], "ltsolve_launches": ui["launches"],
                "lsolve_tasks": li["tasks"], "row_levels": li["row_levels"],
            },
            "pct_hbm_roofline": 100.0 * (value / world) / HBM_PEAK_GBPS,
            "solves_per_s": world * nrhs / (t_max / args.steps),
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "kernel": "solve_level + solve_level_small + solve_clusters: all launches of one lsolve+ltsolve step "
                          "(one 'launch' of the path = one step; HIP events on the launch stream)",
            },
            "setup": setup,
            "check": check,
            "extras": extras,
        }
        out["roofline"]["traffic"] = measured_traffic(out["config"])
        # ---- CPU baseline: the oracle (C restatement of the reference's loops), 1 thread, same factor -------
        if args.cpu_rhs > 0 and world == 1:
            from oracle import pyoracle as orc
            Lo = orc.lib(native=True)
            s = args.cpu_rhs
            Xc = (2.0 * synth.splitmix64(9, n * s) - 1.0)
            t0 = time.perf_counter()
            Lo.orc_solve_llt_many(n, np.ascontiguousarray(Lp, np.int64), np.ascontiguousarray(Li, np.int64),
                                  np.ascontiguousarray(Lx), Xc, n, s)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {
                "value": 2 * bytes_trisolve(n, lnz, 1) * s / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                "sample": f"{s} right-hand sides through lsolve+ltsolve on the same factor ({dt:.1f} s, "
                          f"oracle/rsparse_oracle.c -O3 -march=native -ffp-contract=off, 1 thread)",
                "solves_per_s": s / dt,
            }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
