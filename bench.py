#!/usr/bin/env python3
"""Benchmark of the numeric solve path on MI355X (contract: see the task statement / DESIGN.md "Measurement").

Default workload (BASELINE.json configs[1], "C2"): 2-D 5-point Laplacian 1024 x 1024 (n = 1 048 576), AMD ordering
(order = 0), Cholesky factor L computed by the GPU numeric Cholesky.  One STEP = lsolve + ltsolve of a block of
`--nrhs` right-hand sides per GPU (row-major, resident in HBM), i.e. the solve phase of cholsol applied to a
RHS block.  value = algorithmic GB/s summed over all ranks:
    bytes_step = 2 * (nnz(L) * 12 + (n + 1) * 4 + 16 * n * nrhs)          
Multi-GPU (one process per GPU, torch.distributed/RCCL): rank 0 factors, the factor is broadcast once
(untimed set-up), every rank solves its own RHS block -> weak scaling, no collective in the timed region.

Other modes (parity/bench lines of their own, never the headline):
    --config c3 [--scaling strong]   3-D 7-point Laplacian (default 80^3), 256 right-hand sides; "strong" splits the 256
                                     columns over the ranks (256 / G each) instead of giving every rank 256
    --config c5                      batch of 4 096 independent 2 000-unknown SPD systems through rsp_cholsol_batch_f64
                                     (host arrays in, host arrays out: PCIe inclusive), systems split over the ranks

`python bench.py --gpus N` without a launcher starts the N ranks itself (one child process per GPU, rendezvous on
127.0.0.1) and forwards rank 0's JSON line; under `python -m torch.distributed.run` it takes the ranks it is given and
refuses to run when --gpus disagrees with WORLD_SIZE.
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F64_PEAK_TFLOPS = 78.6  # dense f64 matrix-core peak


def bytes_trisolve(n, nnz, k):
    return nnz * 12 + (n + 1) * 4 + 16 * n * k


# what the triangular-solve kernels and their programmes are built from (the C ABI around them, the Cholesky and the host
# factorisations are not: a change there does not move the solve's traffic)
SOLVE_SOURCES = ("device/blocked_kernels.hip", "device/item_kernels.hip", "device/cluster_kernels.hip", "device/solve_kernels.hip",
                 "device/tri_engine.hip", "device/engine.hpp", "host/blocked.cpp", "host/schedule.cpp",
                 "host/sparse.hpp")


def roofline_of_step(bytes_step, flops_step, seconds):
    """The bound of one lsolve+ltsolve step by the rule roofline_chol uses: max(bytes / HBM peak, flops / f64 matrix-core peak).
    flops = 2 sweeps x 2 x nnz(L) per right-hand side.  C2 at 256 columns is HBM-bound (5 flop/B), C3 is not (32 flop/B)."""
    t_hbm = bytes_step / (HBM_PEAK_GBPS * 1e9)
    t_mfma = flops_step / (MFMA_F64_PEAK_TFLOPS * 1e12)
    kernel = ("solve_level + solve_items + solve_level_small + solve_clusters: all launches of one lsolve+ltsolve step "
              "(one 'launch' of the path = one step; HIP events on the launch stream)")
    if t_hbm >= t_mfma:
        achieved = bytes_step / seconds / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": None, "kernel": kernel}
    achieved = flops_step / seconds / 1e12
    return {"bound": "mfma", "achieved": achieved, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_F64_PEAK_TFLOPS,
            "traffic": None, "kernel": kernel, "flops_per_step": flops_step, "hbm_GBps": bytes_step / seconds / 1e9}


def kernels_sha():
    """Hash of the sources the solve kernels and programmes are built from: a committed counter profile only speaks for
    the code it was measured on."""
    h = hashlib.sha256()
    for f in SOLVE_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(ROOT, "rsparse_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


# the factorisation kernels (chol_engine.* also holds the solve-phase plumbing of a handle, which changes more often and
# does not move the factorisation's traffic)
CHOL_SOURCES = ("device/chol_kernels.hip",)


def chol_sha():
    h = hashlib.sha256()
    for f in CHOL_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(ROOT, "rsparse_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def measured_chol_traffic():
    """HBM bytes of one numeric factorisation at C2 from the committed --pmc passes (profiles/*chol*_summary.json, made by
    tools/collect_chol_profile.py), or None when the profile was measured on other Cholesky sources."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c2_chol_summary.json"))):
        try:
            d = json.load(open(f))
            if d.get("chol_sha") == chol_sha():
                best = (int(d["traffic"]["hbm_bytes_per_factorisation"]), d["tag"])
        except Exception:
            pass
    return best


def measured_traffic(cfg):
    """HBM bytes of one step from the committed rocprofv3 --pmc passes (profiles/*_summary.json, made by
    tools/pmc.sh + tools/collect_profiles.py on this same command), with the profile's tag -- or (None, None) when no
    pass matches this configuration (same matrix, right-hand sides, programme) AND these kernel sources: counters
    cannot be read from inside the timed process, and a profile of older kernels says nothing about these."""
    keys = ("n", "nnz_L", "nrhs_per_gpu", "lsolve_tasks", "lsolve_launches", "ltsolve_launches")
    best, tag, sha = None, None, kernels_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        t, c = d.get("traffic"), d.get("config", {})
        if t and d.get("kernels_sha") == sha and all(c.get(k) == cfg.get(k) for k in keys):
            best, tag = t["hbm_bytes_per_step"], d.get("tag")
    return best, tag


def spawn_ranks(args):
    """No launcher: start args.gpus child processes of this script, one rank each, and forward rank 0's line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    sys.stdout.write("".join(ln + "\n" for ln in out0.splitlines() if ln.startswith("{")))  # the line, not a backend's chatter
    sys.stdout.flush()
    sys.exit(0 if all(c == 0 for c in codes) else 1)


def lu_c4_extras(rs, synth, np):
    """SURVEY 8f item 2 beside the headline figure: numeric LU of BASELINE.json's C4 matrix (banded-random, 200 000 unknowns,
    lusol(order=1, tol=1.0)) by the host routine (the reference's algorithm, one thread) and by the GPU path (patterns on the
    host, values by lu_columns, pivot rule checked on the device), the factors compared BIT FOR BIT; a refactorisation on the
    same pattern; the one-shot lusol and the same call again (analysis cache)."""
    out = {}
    A = rs.Sprs.from_tuple(synth.banded_unsymmetric(200000))
    S = rs.sqr(A, 1)
    old = os.environ.get("RSP_LU_GPU")
    try:
        os.environ["RSP_LU_GPU"] = "0"
        t0 = time.time(); H = rs.lu(A, S, 1.0); out["host_s"] = time.time() - t0
        os.environ.pop("RSP_LU_GPU")
        t0 = time.time(); G = rs.lu(A, S, 1.0); out["gpu_s"] = time.time() - t0
    finally:
        if old is not None:
            os.environ["RSP_LU_GPU"] = old
    out["on_device"] = bool(G.on_device)
    out["bit_identical_to_host_routine"] = bool(
        np.array_equal(H.pinv, G.pinv) and np.array_equal(H.l.p, G.l.p) and np.array_equal(H.l.i, G.l.i) and np.array_equal(H.u.p, G.u.p)
        and np.array_equal(H.u.i, G.u.i) and np.array_equal(H.l.x.view(np.int64), G.l.x.view(np.int64))
        and np.array_equal(H.u.x.view(np.int64), G.u.x.view(np.int64)))
    out["nnz_L"], out["nnz_U"] = int(G.l.p[-1]), int(G.u.p[-1])
    del H
    A2 = rs.Sprs(A.m, A.n, A.p, A.i, A.x * 1.01)
    t0 = time.time(); G.refactor(A2, 1.0); out["refactor_s"] = time.time() - t0
    out["refactor_on_device"] = bool(G.on_device)
    del G
    b = synth.splitmix64(5, A.n)
    rs.cache_clear()
    x = b.copy()
    t0 = time.time(); rs.lusol(A, x, 1, 1.0); out["lusol_one_shot_s"] = time.time() - t0
    t0 = time.time()
    while rs.cache_stats()["pending"] and time.time() - t0 < 120:
        time.sleep(0.01)
    x = b.copy()
    t0 = time.time(); rs.lusol(A2, x, 1, 1.0); out["lusol_cached_s"] = time.time() - t0
    r = rs.gaxpy(A2, x, -b)
    out["lusol_cached_residual_rel"] = float(np.linalg.norm(r) / np.linalg.norm(b))
    rs.cache_clear()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=("c2", "c3", "c5"), default="c2")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--nrhs", type=int, default=256)
    ap.add_argument("--grid", type=int, default=0, help="Laplacian grid side (default: 1024 for c2 = the headline configuration, 80 for c3)")
    ap.add_argument("--systems", type=int, default=4096, help="c5: systems in the batch")
    ap.add_argument("--cpu-rhs", type=int, default=120, help="right-hand sides of the CPU-baseline sample (0 = no CPU baseline)")
    ap.add_argument("--cpu-chol", type=int, default=1, help="1: the CPU baseline also times schol + chol once (oracle, 1 thread)")
    ap.add_argument("--skip-extras", action="store_true", help="skip the single-RHS cholsol timing")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)  # before anything touches torch or HIP
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: refusing to report a line for the wrong size\n")
        sys.exit(2)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    from rsparse_amd import sharding, synth
    rs.lib().rsp_set_device(dev.index)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(t_local):
        if world == 1:
            return t_local, [t_local]
        tt = torch.zeros(world, dtype=torch.float64, device=dev if not rehearsal else "cpu")
        tt[rank] = t_local
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dist.barrier()
        per = [float(v) for v in tt.cpu()]
        return max(per), per

    if args.config == "c5":
        bench_batch(args, rank, world, dev, rs, synth, sharding, barrier, max_over_ranks)
        if world > 1:
            dist.destroy_process_group()
        return

    grid = args.grid or (1024 if args.config == "c2" else 80)
    n = grid * grid if args.config == "c2" else grid ** 3
    total_rhs = args.nrhs
    if args.scaling == "strong":
        col0, nrhs = sharding.column_block(total_rhs, world, rank)
    else:
        col0, nrhs = 0, total_rhs
    setup = {}

    # ---- set-up (untimed): rank 0 orders, analyses and factors on its GPU ------------------------------------
    extras = {}
    roofline_chol = None
    A = None
    if rank == 0:
        t0 = time.time()
        A = synth.laplacian_2d(grid) if args.config == "c2" else synth.laplacian_3d(grid)
        setup["generate_s"] = time.time() - t0
        t0 = time.time()
        S = rs.schol(A, 0)
        setup["schol_host_s"] = time.time() - t0
        t0 = time.time()
        F = rs.CholHandle(A, S)
        setup["chol_analyze_plus_factor_s"] = time.time() - t0
        info = F.info()
        L = F.factor()
        pinv = np.array(S.pinv)
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
            if args.config == "c2":
                # the literal configs[1]: ONE cholsol call from scratch (ordering, analysis, factorisation, solve, PCIe)
                xe = b.copy()
                t0 = time.time()
                rs.cholsol(A, xe, 0)
                extras["cholsol_one_shot_s"] = time.time() - t0
                extras["cholsol_one_shot_x_error"] = float(np.max(np.abs(xe - xstar)) / np.max(np.abs(xstar)))
                # ... and the same call again once the library's background analysis of that pattern is in its cache
                # (rsp_cholsol_f64's analysis cache: refactorisation + solve on the cached handle, nothing else)
                t0 = time.time()
                while rs.cache_stats()["pending"] and time.time() - t0 < 120:
                    time.sleep(0.01)
                extras["cholsol_cache_fill_wait_s"] = time.time() - t0
                for key in ("cholsol_cached_first_hit_s", "cholsol_cached_s"):
                    xe = b.copy()
                    t0 = time.time()
                    rs.cholsol(A, xe, 0)
                    extras[key] = time.time() - t0
                extras["cholsol_cached_x_error"] = float(np.max(np.abs(xe - xstar)) / np.max(np.abs(xstar)))
                extras["cholsol_cache"] = rs.cache_stats()
                rs.cache_clear()
        if not args.skip_extras and args.config == "c2" and world == 1:
            extras["lu_c4"] = lu_c4_extras(rs, synth, np)
        extras["chol_factor_first_call_ms"] = F.info()["factor_ms"]  # includes first-launch costs of every kernel
        if not args.skip_extras:
            # a time step of an implicit solver: new values on the same pattern, then a solve.  The resident solve programmes
            # take the new factor's values on the device (refill kernels); the first round also sends their recipe there.
            xr = torch.tensor(b, device=dev)
            for rnd in range(2):
                t0 = time.time()
                F.refactor(A)
                t1 = time.time()
                F.solve_dev(xr.data_ptr(), rs.api.COL_MAJOR, n, 1, True, torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                t2 = time.time()
                extras["refactor_call_ms" if rnd else "refactor_call_first_ms"] = (t1 - t0) * 1e3
                extras["solve_after_refactor_ms" if rnd else "solve_after_refactor_first_ms"] = (t2 - t1) * 1e3
        F.refactor(A)  # the figure that is priced: numeric factorisation on the resident analysis (rsp_chol_refactor_f64)
        fi = F.info()
        extras["chol_factor_ms"] = fi["factor_ms"]
        extras["chol_flops"] = fi["flops"]
        extras["chol_bytes"] = fi["bytes_chol"]
        extras["chol_levels"] = fi["chol_levels"]
        extras["supernodes"] = fi["supernodes"]
        # the numeric Cholesky against ITS bound: the larger of its algorithmic bytes at HBM speed and its flops at the
        # dense f64 matrix-core peak (HIP events around the factorisation on the library's stream, rsp_chol_factor_ms)
        t_hbm = fi["bytes_chol"] / (HBM_PEAK_GBPS * 1e9)
        t_mfma = fi["flops"] / (MFMA_F64_PEAK_TFLOPS * 1e12)
        t_meas = fi["factor_ms"] * 1e-3
        if t_mfma >= t_hbm:
            roofline_chol = {"bound": "mfma", "achieved": fi["flops"] / t_meas / 1e12, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s"}
        else:
            roofline_chol = {"bound": "hbm", "achieved": fi["bytes_chol"] / t_meas / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
        chol_traffic = measured_chol_traffic() if args.config == "c2" else None
        roofline_chol.update({"frac": roofline_chol["achieved"] / roofline_chol["peak"], "traffic": chol_traffic[0] if chol_traffic else None,
                              "traffic_profile": chol_traffic[1] if chol_traffic else None, "ms": fi["factor_ms"],
                              "bytes": fi["bytes_chol"], "flops": fi["flops"],
                              "kernel": "chol_panel_factor + chol_tile_update + chol_tiny_panels: one numeric factorisation"})
        del F
        Lp, Li, Lx = L.p, L.i.astype(np.int64), L.x

    # ---- the factor travels to the other GPUs once (RCCL broadcast over xGMI), untimed -------------------------
    if world > 1:
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

    # ---- RHS block of this rank, resident in HBM: SplitMix64 block of seed 2 + rank, mapped to [-1, 1) -----------
    have = nrhs > 0
    X = synth.splitmix64_torch(2 + rank, n * max(nrhs, 1), dev, -1.0, 1.0).reshape(n, max(nrhs, 1))
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        # lsolve then ltsolve on the same block; the library pipelines column parts of the block over several streams
        if have:
            rs.solve_pair_dev(lower, upper, X.data_ptr(), nrhs, nrhs, stream)

    # correctness of the timed path itself (rank 0): EVERY column's residual of (P A P') y = b, computed on the device
    # by rsp_mat_gaxpy_dev_f64 (L L' = P A P': what the solve phase of cholsol inverts), plus L L' y = b on column 0
    # through scipy's products
    check = {}
    if rank == 0 and have:
        import scipy.sparse as sp
        B0 = X.clone()
        step()
        torch.cuda.synchronize()
        Ap = sp.csc_matrix((A[4], (pinv[A[3]], np.repeat(pinv, np.diff(A[2])))), shape=(n, n)).tocsc()
        Ap.sort_indices()
        M = rs.MatHandle((n, n, Ap.indptr.astype(np.int64), Ap.indices.astype(np.int64), Ap.data))
        R = B0.clone()
        M.gaxpy_dev(X.data_ptr(), nrhs, R.data_ptr(), nrhs, nrhs, rs.api.ROW_MAJOR, -1.0, stream)
        torch.cuda.synchronize()
        rel = R.norm(dim=0) / B0.norm(dim=0)
        check["residual_max_all_columns"] = float(rel.max())
        check["columns_checked"] = int(nrhs)
        spread = sorted(set(int(c) for c in np.linspace(0, nrhs - 1, 8)))
        check["residual_spread_columns"] = {str(c): float(rel[c]) for c in spread}
        y = X[:, 0].cpu().numpy()
        b0 = B0[:, 0].cpu().numpy()
        Lsp = sp.csc_matrix((Lx, Li, Lp), shape=(n, n))
        r = Lsp @ (Lsp.T @ y) - b0
        check["LLt_residual_col0"] = float(np.linalg.norm(r) / np.linalg.norm(b0))
        del M, R, B0, rel
    else:
        step()
    for _ in range(max(0, args.warmup - 1)):
        step()

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
    t_max, t_per_rank = max_over_ranks(t_local)

    if rank == 0:
        counts = [sharding.column_block(total_rhs, world, r)[1] for r in range(world)] if args.scaling == "strong" else [nrhs] * world
        bytes_rank = [2 * bytes_trisolve(n, lnz, c) if c else 0 for c in counts]
        bytes_step = bytes_rank[0]
        ms_per_step = t_max * 1e3 / args.steps
        value = sum(bytes_rank) / (t_max / args.steps) / 1e9
        achieved = bytes_step / (ev_ms * 1e-3 / args.steps) / 1e9  # HIP events on the launch stream, this rank
        li, ui = lower.info(), upper.info()
        name = (f"C2: 2D 5-point Laplacian {grid}x{grid}" if args.config == "c2" else f"C3: 3D 7-point Laplacian {grid}^3")
        out = {
            "metric": "triangular-solve GB/s",
            "value": value,
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{name}, cholsol(order=0) solve phase (lsolve+ltsolve) on "
                            + (f"{total_rhs} RHS split over the GPUs" if args.scaling == "strong" else f"{nrhs} RHS per GPU")
                            + ", factor from the GPU numeric Cholesky",
                "n": n, "nnz_L": lnz, "nrhs_per_gpu": nrhs, "rhs_layout": "row-major",
                "bytes_per_step": bytes_step,
                "lsolve_launches": li["launches"], "ltsolve_launches": ui["launches"],
                "lsolve_tasks": li["tasks"], "row_levels": li["row_levels"],
            },
            "pct_hbm_roofline": 100.0 * (value / world) / HBM_PEAK_GBPS,
            "solves_per_s": sum(counts) / (t_max / args.steps),
            "roofline": roofline_of_step(bytes_step, 4.0 * lnz * counts[0], ev_ms * 1e-3 / args.steps),
            "setup": setup,
            "check": check,
            "extras": extras,
        }
        if world > 1:
            out["ms_per_step_per_rank"] = [t * 1e3 / args.steps for t in t_per_rank]
        if args.scaling == "strong":
            # what a narrower block costs: the coefficients are read once per 64 columns whatever the block's width
            out["config"]["rhs_total"] = total_rhs
            out["config"]["columns_per_rank"] = counts
            out["strong_scaling_note"] = (f"{total_rhs} columns over {world} ranks = {counts[0]} per rank; every rank still reads the whole "
                                          f"factor ({lnz * 12 / 1e9:.2f} GB per sweep), so efficiency falls as columns per rank fall")
        traffic, tag = measured_traffic(out["config"])
        out["roofline"]["traffic"] = traffic
        out["roofline"]["traffic_profile"] = tag
        out["roofline"]["kernels_sha"] = kernels_sha()
        if roofline_chol is not None:
            out["roofline_chol"] = roofline_chol
        # ---- CPU baseline: the oracle (C restatement of the reference's loops), 1 thread, same factor -------
        if args.cpu_rhs > 0:
            from oracle import pyoracle as orc
            Lo = orc.lib(native=True)
            s = args.cpu_rhs
            Xc = (2.0 * synth.splitmix64(9, n * s) - 1.0)
            t0 = time.perf_counter()
            Lo.orc_solve_llt_many(n, np.ascontiguousarray(Lp, np.int64), np.ascontiguousarray(Li, np.int64),
                                  np.ascontiguousarray(Lx), Xc, n, s)
            dt = time.perf_counter() - t0
            try:
                affinity = len(os.sched_getaffinity(0))
            except AttributeError:
                affinity = None
            out["cpu_baseline"] = {
                "value": 2 * bytes_trisolve(n, lnz, 1) * s / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                "sample": f"{s} right-hand sides through lsolve+ltsolve on the same factor ({dt:.1f} s, "
                          f"oracle/rsparse_oracle.c -O3 -march=native -ffp-contract=off, 1 thread, rank 0 only)",
                "solves_per_s": s / dt,
                "nproc": os.cpu_count(), "cpus_allowed": affinity,
            }
            if args.cpu_chol and A is not None:
                t0 = time.perf_counter()
                So = orc.schol(A, 0)
                t1 = time.perf_counter()
                st, _ = orc.chol(A, So)
                t2 = time.perf_counter()
                out["cpu_baseline"]["schol_s"] = t1 - t0
                out["cpu_baseline"]["chol_s"] = t2 - t1 if st == orc.OK else None
                out["cpu_baseline"]["chol_note"] = "oracle schol (ordering + symbolic) and up-looking chol once, 1 thread, same matrix"
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def bench_batch(args, rank, world, dev, rs, synth, sharding, barrier, max_over_ranks):
    """--config c5: 4 096 (or --systems) independent 2 000-unknown SPD systems that share one pattern; every rank takes a
    contiguous range of systems through rsp_cholsol_batch_f64 (host arrays in and out: the figure is PCIe inclusive)."""
    n, nsys = 2000, args.systems
    s0, cnt = sharding.column_block(nsys, world, rank)
    A0 = synth.circuit_spd(n, 6, 100)
    nnz = len(A0[4])
    vals = np.empty((max(cnt, 1), nnz))
    B = np.empty((max(cnt, 1), n))
    for k in range(cnt):
        vals[k] = synth.circuit_spd(n, 6, 100 + s0 + k)[4]
        B[k] = synth.splitmix64(1000 + s0 + k, n)
    work = B.copy()

    def step():
        work[:] = B
        if cnt:
            st = rs.cholsol_batch(A0, vals[:cnt], work[:cnt], 0)
            assert not st.any()

    for _ in range(max(1, args.warmup)):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_local = time.perf_counter() - t0
    t_max, per = max_over_ranks(t_local)
    if rank == 0:
        import scipy.sparse as sp
        worst = 0.0
        for k in range(0, cnt, max(1, cnt // 64)):
            Asp = sp.csc_matrix((vals[k], A0[3], A0[2]), shape=(n, n))
            worst = max(worst, float(np.linalg.norm(Asp @ work[k] - B[k]) / np.linalg.norm(B[k])))
        out = {
            "metric": "cholsol systems/s (batched)", "value": nsys / (t_max / args.steps), "unit": "systems/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_max * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C5: {nsys} independent circuit-style SPD systems of {n} unknowns, one pattern, cholsol(order=0) "
                                   f"through rsp_cholsol_batch_f64, host arrays (PCIe inclusive), systems split over the GPUs",
                       "n": n, "systems": nsys, "systems_rank0": cnt, "nnz_A": nnz},
            "check": {"residual_max_sampled": worst},
        }
        if world > 1:
            out["ms_per_step_per_rank"] = [t * 1e3 / args.steps for t in per]
        print(json.dumps(out))


if __name__ == "__main__":
    main()
