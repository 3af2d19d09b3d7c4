#!/usr/bin/env python3
"""bench.py -- CCD++ outer-iteration throughput on MI355X (BASELINE.json metric:
"rating-nnz/sec per CCD++ outer iter at k=64").

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: starts N worker processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is ONE full CCD++ outer iteration (all k ranks: subtract + add-back + v-sweep + u-sweep
for every rank, T = 1) over a synthetic Netflix-shaped rating matrix that is generated in HBM
(SURVEY.md 8d workload (i): 480189 x 17770, Z = 99 072 112, k = 64, lambda = 0.05).  With N > 1
every rank owns one user-row block of that size (weak scaling: the global matrix has N x the
rows and N x the non-zeros) and the ranks exchange one RCCL all-reduce of the (g, h) column
partials per inner iteration.  Inputs are resident in HBM when the timed region starts.

--workload config5 switches to BASELINE configs[4]: ONE global 10M x 1M matrix with 1e9 ratings, k = 128,
row-sharded over the N ranks (strong scaling: every rank draws the same matrix from the same seed and
keeps its nnz-balanced row block), 8 MB all-reduce per inner iteration.

The JSON line also carries
  roofline      the dominant kernel's algorithmic bytes / its mean launch time (HIP events on the
                solver's own stream, second pass over the same K steps) against 8 TB/s
  cpu_baseline  the CPU oracle (bit-exact restatement of the reference's ccdr1_OMP) timed on this
                box's host cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "cuda-recommender_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def host_cores() -> int:
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except Exception:
            continue
    return max(1, n)


def als_measure(a, d, mfx, k, steps, warmup) -> dict:
    """ALS iterations (BASELINE configs[3]) on the device-resident matrix `d`: time per iteration, the two half-sweeps
    against both of their rooflines, test RMSE."""
    import time as _t
    import torch
    p = mfx.parameter()
    p.k, p.lambda_ = k, a.lam
    s = mfx.AlsSolver(None, None, p, device_arrays=d)
    rows, cols, Z = int(d["rows"]), int(d["cols"]), int(d["csr_val"].numel())
    s.set_factors(mfx.initial_col(cols, k))
    s.iterate(warmup, with_rmse=False)
    torch.cuda.synchronize()
    t0 = _t.perf_counter()
    rep = s.iterate(steps, with_rmse=True)
    el = _t.perf_counter() - t0
    kt = s.kernel_times()
    s.close()
    flops = 2.0 * (Z * k * (k + 1) + 2.0 * Z * k)  # both half-sweeps: symmetric Gramian + rhs
    # Per half-sweep: the matrix-core roofline (Z k (k + 1) + 2 Z k flop of symmetric Gramian + rhs) AND the gather
    # roofline -- SURVEY 8d: Z (8 + 4 k) bytes per half-sweep (index + rating + one k-float factor row per rating).  The
    # factor rows come from L2 / Infinity Cache rather than HBM (the table is 4.5 / 123 MB), so the HBM peak is the
    # generous yardstick; MI355X_MICROARCH.md puts the row-gather ceiling at 5.5-5.8 TB/s.
    halves = {}
    for name, (secs, n) in kt.items():
        if n:
            t_half = secs / n
            halves[name] = {"ms": round(1e3 * t_half, 3),
                            "mfma": {"achieved": round(0.5 * flops / t_half / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s",
                                     "frac": round(0.5 * flops / t_half / 1e12 / 157.3, 4)},
                            "gather": {"algorithmic_bytes": int(Z * (8 + 4 * k)), "achieved": round(Z * (8 + 4 * k) / t_half / 1e9, 1),
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(Z * (8 + 4 * k) / t_half / 1e9 / HBM_PEAK_GBS, 4)}}
    traffic, traffic_source = None, "none: no PMC record for the ALS half-sweeps at this size in profiles/traffic.json"
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        ent = tj.get(f"als_iteration_k{k}@{Z}")
        if ent and ent.get("kernel_src_sha16") == als_source_hash():
            traffic = ent.get("hbm_bytes_per_launch")
            traffic_source = f"profiles/traffic.json: rocprofv3 --pmc passes of {ent.get('collected', '?')} (both half-sweeps of one iteration)"
        elif ent:
            traffic_source = "stale: als_solver.hip changed since profiles/traffic.json was collected"
    except Exception:
        pass
    ms = 1e3 * el / steps
    return {"ms_per_iteration": round(ms, 3), "k": k, "steps": steps, "warmup": warmup, "workload": f"{rows}x{cols} nnz={Z} k={k}",
            "gramian_tflops": round(flops / (el / steps) / 1e12, 2),
            # whole iteration (Gramians + 2 x nseg Cholesky/solves) against the fp32 MFMA peak,
            # counting only the symmetric half of each Gramian as useful work
            "roofline": {"bound": "mfma", "achieved": round(flops / (el / steps) / 1e12, 2), "peak": 157.3,
                         "unit": "TFLOP/s", "frac": round(flops / (el / steps) / 1e12 / 157.3, 4), "traffic": traffic,
                         "traffic_source": traffic_source},
            "half_sweeps": halves,
            "kernels": {n: {"total_ms": round(v[0] * 1e3, 3), "launches": int(v[1])} for n, v in kt.items()},
            "rmse": [round(r.rmse, 6) for r in rep], "als_src_sha16": als_source_hash()}


def bench_als(a, d, mfx, gen_s) -> None:
    """Secondary measurement (BASELINE configs[3]): ALS iteration time on the same synthetic matrix, as its own line."""
    m = als_measure(a, d, mfx, a.k, a.steps, a.warmup)
    print(json.dumps({"metric": "ALS iteration time at k=%d" % a.k, "value": m["ms_per_iteration"], "unit": "ms",
                      "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "higher_is_better": False, "dtype": "f32",
                      "data": "synthetic", "config": {"workload": m["workload"]},
                      "gramian_tflops": m["gramian_tflops"], "roofline": m["roofline"], "half_sweeps": m["half_sweeps"],
                      "kernels": m["kernels"], "rmse": m["rmse"], "als_src_sha16": m["als_src_sha16"],
                      "gen_seconds": round(gen_s, 2)}), flush=True)


def kernel_source_hash() -> str:
    """sha256 (first 16 hex digits) over the CCD++ kernel sources: what ties a PMC record in profiles/traffic.json
    to the kernels it was measured on (tools/collect_profiles.py writes the same value)."""
    import hashlib
    h = hashlib.sha256()
    for f in CCD_HASHED_SOURCES:
        h.update(open(os.path.join(ROOT, "cuda-recommender_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


# (layout choices -- panel counts, workgroup ranges -- live in ccd_solver.hip / layout_kernels.hip and move the traffic too)
CCD_HASHED_SOURCES = ("ccd_kernels.hip", "ccd_scatter.hip", "flat_layout.hpp", "ccd_solver.hip", "layout_kernels.hip")


def als_source_hash() -> str:
    import hashlib
    return hashlib.sha256(open(os.path.join(ROOT, "cuda-recommender_amd", "csrc", "als_solver.hip"), "rb").read()).hexdigest()[:16]


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_workers(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh worker processes of this script (one
    per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would), relay rank 0's
    stdout (the single JSON line), exit with the workers' status.  Runs before anything touches the GPU.
    ALL children are supervised: the first one that exits non-zero (an out-of-memory kill while the matrix is
    generated, a failed setup) ends the run -- the others, which would otherwise sit in a collective waiting
    for it until the driver's timeout, are terminated (the exact children started here, nothing else), and
    that status is returned with a one-line diagnosis on stderr."""
    import subprocess
    import tempfile
    env0 = dict(os.environ)
    env0.update({"WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()),
                 "LOCAL_WORLD_SIZE": str(n)})
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out0 = tempfile.TemporaryFile()
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rc, live = 0, set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = code
                print(f"[bench launcher] rank {r} exited with status {code}; stopping the other ranks", file=sys.stderr, flush=True)
                break
        else:
            time.sleep(0.2)
    for r in sorted(live):  # only after a failure: the survivors are waiting for a rank that is gone
        procs[r].terminate()
    for r in sorted(live):
        try:
            procs[r].wait(timeout=10)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    out0.seek(0)
    sys.stdout.write(out0.read().decode(errors="replace"))
    sys.stdout.flush()
    return rc


def dry_run(a) -> None:
    """Rehearsal of the launch plumbing without a GPU: the ranks meet over gloo, all-reduce their rank
    numbers and rank 0 prints one JSON line.  (tests/test_host.py runs it with 2 processes.)"""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.dry_run_fail_rank == rank:  # (test hook) this rank dies before it joins the collective
        raise SystemExit(3)
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.init_process_group("gloo")
        dist.all_reduce(t)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out = {"dry_run": True, "n_gpus": world, "rank_sum": float(t[0]), "workload": a.workload}
        if world > 1 and a.workload == "netflix" and not a.no_strong:  # what the real run adds at N > 1 (same keys, no values)
            out["config5_strong"] = {k: None for k in CONFIG5_STRONG_KEYS}
        print(json.dumps(out), flush=True)


CONFIG5_STRONG_KEYS = ("workload", "scaling", "k", "steps", "warmup", "ms_per_step", "value", "unit", "overlap_groups", "by_overlap_groups",
                       "allreduce_us_per_inner_iter", "host_enqueue_ms_per_step", "rank_nnz_min", "rank_nnz_max", "speedup_vs_n1",
                       "n1_ms_per_step", "n1_source", "layout", "gen_seconds")


def config5_strong_leg(a, torch, dist, mfx, synth_torch, comm, world, rank, local_rank, dev) -> dict:
    """(r4) The north star's OWN scaling workload on the ranks of an N > 1 run, next to the weak-scaling headline: BASELINE
    configs[4] -- ONE global 10 M x 1 M matrix with 1e9 ratings, k = 128, nnz-balanced user-row blocks, one 8 MB all-reduce
    of the column partials per inner iteration -- timed for a.strong_steps outer iterations per setting of the
    panel-group overlap (MFX_OVERLAP_GROUPS = 1: the exchange fully exposed; 2: the first half of the columns is
    exchanged under the second half's pass).  The better one is the leg's value; both are listed.  speedup_vs_n1 is against
    the committed single-GPU record of the same workload (profiles/, named in n1_source)."""
    t0 = time.time()
    rows, cols, nnz, k = 10_000_000, 1_000_000, 1_000_000_000, 128
    d = synth_torch.synth_ratings_device(rows, cols, nnz, seed=a.seed, device=dev, sigma_rows=0.5, sigma_cols=1.0, shard=(rank, world))
    nnz_local = int(d["csr_val"].numel())
    col_cnt = (d["csc_col_ptr"][1:] - d["csc_col_ptr"][:-1]).to(torch.int32).contiguous()
    tot = torch.tensor([nnz_local, int(d["test_val"].numel())], dtype=torch.int64, device=dev)
    lo, hi = tot[:1].clone(), tot[:1].clone()
    dist.all_reduce(col_cnt); dist.all_reduce(tot)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    W0 = mfx.initial_col(k, int(d["rows"]))
    by, layout = {}, None
    saved = os.environ.get("MFX_OVERLAP_GROUPS")
    for groups in (1, 2):
        os.environ["MFX_OVERLAP_GROUPS"] = str(groups)
        p = mfx.parameter()
        p.k, p.lambda_, p.maxinneriter, p.device = k, a.lam, 1, local_rank
        solver, status = None, 0
        try:
            solver = mfx.CcdSolver(None, None, p, comm=comm, global_col_nnz=col_cnt, global_test_nnz=int(tot[1]), device_arrays=d)
        except mfx.MfxError as ex:
            status = -1
            print(f"[rank {rank}] config5_strong: solver creation failed: {ex}", file=sys.stderr, flush=True)
        if comm.agree(status) != 0:
            raise SystemExit(f"[rank {rank}] config5_strong: a rank failed during setup")
        layout = solver.layout_info()
        solver.set_factors(W0)
        solver.iterate(1, with_rmse=False)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        solver.iterate(a.strong_steps, with_rmse=False)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        te = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        ms = 1e3 * float(te[0]) / a.strong_steps
        solver.set_profile(True)
        solver.iterate(1, with_rmse=False)
        kt = solver.kernel_times()
        solver.close()
        avg = lambda n: round(kt[n][0] / max(1, kt[n][1]) * 1e6, 2) if n in kt else None
        by[str(groups)] = {"ms_per_step": round(ms, 3), "allreduce_us": avg("rccl_allreduce"), "allreduce_calls_per_step": int(kt["rccl_allreduce"][1]) if "rccl_allreduce" in kt else 0,
                           "v_pass_us": avg("ccd_scatter_v_pass"), "u_pass_us": avg("ccd_scatter_u_pass"),
                           "host_enqueue_ms_per_step": round(kt["host_enqueue_outer_iteration"][0] * 1e3, 3) if "host_enqueue_outer_iteration" in kt else None}
    if saved is None:
        os.environ.pop("MFX_OVERLAP_GROUPS", None)
    else:
        os.environ["MFX_OVERLAP_GROUPS"] = saved
    best = min(by, key=lambda g: by[g]["ms_per_step"])
    ms = by[best]["ms_per_step"]
    n1_ms, n1_src = None, "none: no committed single-GPU record of this workload under profiles/"
    for f in ("r04_bench_config5.json", "r03_bench_config5.json"):
        try:
            n1_ms = float(json.load(open(os.path.join(ROOT, "profiles", f)))["ms_per_step"])
            n1_src = f"profiles/{f} (bench.py --workload config5 --gpus 1, builder-run)"
            break
        except Exception:
            continue
    return {"workload": f"synthetic (BASELINE configs[4]): ONE global {rows}x{cols}, nnz={int(tot[0])}, k={k}, T=1, nnz-balanced user-row blocks over {world} GPUs",
            "scaling": "strong", "k": k, "steps": a.strong_steps, "warmup": 1, "ms_per_step": ms, "value": round(int(tot[0]) / (ms * 1e-3), 1), "unit": "nnz/s",
            "overlap_groups": int(best), "by_overlap_groups": by,
            "allreduce_us_per_inner_iter": by[best]["allreduce_us"] * by[best]["allreduce_calls_per_step"] / k if by[best]["allreduce_us"] is not None else None,
            "host_enqueue_ms_per_step": by[best]["host_enqueue_ms_per_step"],
            "rank_nnz_min": int(lo[0]), "rank_nnz_max": int(hi[0]),
            "speedup_vs_n1": round(n1_ms / ms, 3) if n1_ms else None, "n1_ms_per_step": n1_ms, "n1_source": n1_src,
            "layout": layout, "gen_seconds": round(gen_s, 2)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=480189)
    ap.add_argument("--cols", type=int, default=17770)
    ap.add_argument("--nnz", type=int, default=99072112)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--lam", type=float, default=0.05)
    ap.add_argument("--inner", type=int, default=1)
    ap.add_argument("--schedule", type=int, default=1)
    ap.add_argument("--variant", type=int, default=1)
    ap.add_argument("--tiles", type=int, default=0)
    ap.add_argument("--panel-rows", type=int, default=0, help="LDS panel size (0 auto, -1 off)")
    ap.add_argument("--wg-waves", type=int, default=0, help="waves per workgroup of the panel kernel (0 = 16)")
    ap.add_argument("--layout-build", type=int, default=0, help="0 auto (GPU when possible), 1 host, 2 GPU or fail")
    ap.add_argument("--graph", type=int, default=0, help="0 = hipGraph replay of outer iterations, -1 = eager launches")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--sigma-rows", type=float, default=None, help="log-normal sigma of the user activity (generator)")
    ap.add_argument("--sigma-cols", type=float, default=None, help="log-normal sigma of the item popularity (generator)")
    ap.add_argument("--force-comm", action="store_true",
                    help="N = 1 only: run through the sharded code path with a 1-rank RCCL communicator")
    ap.add_argument("--solver", choices=["ccd", "als"], default="ccd", help="als: report ALS iteration time instead")
    ap.add_argument("--no-event-pass", action="store_true",
                    help="skip the second, event-bracketed pass (use under rocprofv3 --kernel-trace: the "
                         "event packets between launches otherwise end up inside its kernel durations)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-als", action="store_true", help="skip the ALS leg (configs[3], 3 iterations at k = 64 on the resident matrix; N = 1, default workload only)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1, default workload: skip the config5_strong leg (configs[4] on the same ranks)")
    ap.add_argument("--force-strong", action="store_true",
                    help="N = 1 only (rehearsal on a one-GPU box): run the config5_strong leg too, through a 1-rank process group and a 1-rank RCCL communicator")
    ap.add_argument("--strong-steps", type=int, default=2, help="timed outer iterations of the config5_strong leg, per overlap setting")
    ap.add_argument("--no-rank-one", action="store_true",
                    help="skip the standalone rank-one sweep measurement (one extra outer iteration at T = 2, N = 1 only)")
    ap.add_argument("--cpu-ranks", type=int, default=16, help="ranks the CPU baseline times (scaled to k): 16 of 64 = ~1 s per outer iteration on 16 cores, two iterations")
    ap.add_argument("--workload", choices=["netflix", "config5", "nnz1e9"], default="netflix",
                    help="netflix: BASELINE configs[2], one 480189-row block per GPU (weak scaling, the metric's "
                         "config); config5: BASELINE configs[4], one global 10M x 1M x 1e9 matrix at k = 128 "
                         "row-sharded over the ranks (strong scaling); nnz1e9: the north star's target point "
                         "'k = 64, nnz = 1e9' on one GPU -- ten Netflix-sized user blocks, 4801890 x 17770, 990721120 ratings")
    ap.add_argument("--dry-run", action="store_true", help="launch plumbing only (gloo, no GPU)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help="(with --dry-run) this rank exits with status 3 before the collective")
    a = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:  # bare `python bench.py --gpus N`: be the launcher
        raise SystemExit(launch_workers(a.gpus, sys.argv[1:]))
    if a.dry_run:
        return dry_run(a)
    given = {x.split("=")[0] for x in sys.argv[1:] if x.startswith("--")}
    if a.workload == "config5":  # shape defaults of configs[4] unless given explicitly
        if "--rows" not in given: a.rows = 10_000_000
        if "--cols" not in given: a.cols = 1_000_000
        if "--nnz" not in given: a.nnz = 1_000_000_000
        if "--k" not in given: a.k = 128
    if a.workload == "nnz1e9":
        if "--rows" not in given: a.rows = 4_801_890
        if "--nnz" not in given: a.nnz = 990_721_120

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device(dev))

    import mfx  # after torch: one HIP runtime / one RCCL in the process
    from mfx import synth_torch
    if mfx.device_count() < 1:
        raise SystemExit("bench.py needs a GPU; libmfx has no CPU path: " + mfx.lib().mfx_last_error().decode())

    # ---------------- synthetic input, generated in HBM ----------------
    t0 = time.time()
    strong = a.workload == "config5"
    d_sigma = ((0.5 if a.sigma_rows is None else a.sigma_rows, 1.0 if a.sigma_cols is None else a.sigma_cols) if strong else
               (1.2 if a.sigma_rows is None else a.sigma_rows, 1.8 if a.sigma_cols is None else a.sigma_cols))
    if strong:
        # every rank draws the SAME global matrix and keeps its nnz-balanced block of user rows;
        # uniform user activity / mild item skew: config 5 is "synthetic 10M x 1M", not a Netflix-shaped one
        d = synth_torch.synth_ratings_device(a.rows, a.cols, a.nnz, seed=a.seed, device=dev,
                                             sigma_rows=0.5 if a.sigma_rows is None else a.sigma_rows,
                                             sigma_cols=1.0 if a.sigma_cols is None else a.sigma_cols, shard=(rank, world))
    else:
        d = synth_torch.synth_ratings_device(a.rows, a.cols, a.nnz, seed=a.seed + 7919 * rank, device=dev,
                                             sigma_rows=1.2 if a.sigma_rows is None else a.sigma_rows,
                                             sigma_cols=1.8 if a.sigma_cols is None else a.sigma_cols)
    nnz_local = int(d["csr_val"].numel())
    col_cnt = (d["csc_col_ptr"][1:] - d["csc_col_ptr"][:-1]).to(torch.int32).contiguous()
    nnz_tot = torch.tensor([nnz_local, int(d["test_val"].numel())], dtype=torch.int64, device=dev)
    comm = None
    if world > 1:
        dist.all_reduce(col_cnt)
        dist.all_reduce(nnz_tot)
        uid = [mfx.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = mfx.Comm(uid[0], rank, world, local_rank)
    if world == 1 and a.force_comm:
        comm = mfx.Comm(mfx.Comm.unique_id(), 0, 1, local_rank)
    nnz_global, ntest_global = int(nnz_tot[0]), int(nnz_tot[1])
    torch.cuda.synchronize()
    gen_s = time.time() - t0

    if a.solver == "als":
        return bench_als(a, d, mfx, gen_s)
    p = mfx.parameter()
    p.k, p.lambda_, p.maxinneriter, p.device = a.k, a.lam, a.inner, local_rank
    p.schedule, p.kernel_variant, p.tiles_per_span = a.schedule, a.variant, a.tiles
    p.panel_rows, p.wg_waves, p.graph, p.layout_build = a.panel_rows, a.wg_waves, a.graph, a.layout_build
    t0 = time.time()
    solver, status = None, 0
    try:
        solver = mfx.CcdSolver(None, None, p, comm=comm, global_col_nnz=col_cnt if comm else None,
                               global_test_nnz=ntest_global, device_arrays=d)
    except mfx.MfxError as ex:
        if comm is None:
            raise
        status = -1
        print(f"[rank {rank}] solver creation failed: {ex}", file=sys.stderr, flush=True)
    if comm is not None:  # nobody enters a collective unless everybody's setup succeeded
        worst = comm.agree(status)
        if worst != 0:
            raise SystemExit(f"[rank {rank}] a rank failed during setup (status {worst})")
    setup_s = time.time() - t0  # one-time: panel layout build (+ RCCL connection setup when sharded)
    layout = solver.layout_info()
    W0 = mfx.initial_col(a.k, int(d["rows"]))  # reference init (glibc rand, seed 0), src/tools.cpp:165-173
    solver.set_factors(W0)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---------------- warmup, then EXACTLY K timed steps ----------------
    if a.warmup > 0:
        solver.iterate(a.warmup, with_rmse=False)
    sync()
    t0 = time.perf_counter()
    solver.iterate(a.steps, with_rmse=False)  # blocks until the solver's stream has drained
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te[0])
    ms_per_step = 1e3 * elapsed / a.steps
    value = nnz_global / (elapsed / a.steps)

    # ---------------- roofline: same K steps again, every launch bracketed by HIP events ------------
    ktimes = {}
    if not a.no_event_pass:
        solver.set_profile(True)
        solver.iterate(a.steps, with_rmse=False)
        ktimes = solver.kernel_times()
        solver.set_profile(False)
    m, n, Z = int(d["rows"]), int(d["cols"]), nnz_local
    flags = Z / 8.0
    # algorithmic bytes per launch (DESIGN.md "bytes per kernel"): idx + val read + val write per
    # non-zero, head flags, one compulsory read of each operand pack, the partial-sum outputs
    alg = {
        "ccd_fused_csc_pass": 12.0 * Z + flags + 8.0 * m + 8.0 * n + 8.0 * n,
        "ccd_fused_csr_pass": 12.0 * Z + flags + 16.0 * n + 8.0 * m + 8.0 * m,
        "ccd_flat_sweep": 8.0 * Z + flags + 4.0 * max(m, n) + 8.0 * min(m, n),
        "ccd_flat_resid": 12.0 * Z + flags + 4.0 * (m + n),
        "ccd_wave_sweep": 8.0 * Z + 4.0 * (m + n) + 8.0 * min(m, n),
        "ccd_wave_resid": 12.0 * Z + 4.0 * (m + n),
        "ccd_ref_order_sweep": 8.0 * Z + 4.0 * (m + n) + 8.0 * min(m, n),  # parity mode (kernel_variant -1): latency-, not bandwidth-bound
        # scatter layout (hyper-sparse shards): the same contract as the fused passes they replace
        "ccd_scatter_v_pass": 12.0 * Z + flags + 8.0 * m + 8.0 * n + 8.0 * n,
        "ccd_scatter_u_pass": 12.0 * Z + flags + 16.0 * n + 8.0 * m + 8.0 * m,
        "ccd_scatter_sweep": 8.0 * Z + flags + 4.0 * max(m, n) + 8.0 * min(m, n),
    }
    if a.variant == -1 and os.environ.get("MFX_REF_FUSED", "1") != "0":
        # (r4) the mode's launches are owner passes on the default schedule: per rank two fused passes (the fused passes' contract,
        # mean of the two copies) and 2 (T - 1) read-only sweeps
        fused = 12.0 * Z + 12.0 * (m + n) + 8.0 * (m + n)
        alg["ccd_ref_order_sweep"] = (fused + (a.inner - 1) * alg["ccd_ref_order_sweep"]) / a.inner
    # bytes the LDS-panel kernels physically stream per non-zero: 16-bit local index + fp32 value read
    # + fp32 value written (the contract figure above keeps SURVEY 8d's 32-bit index)
    phys = {"ccd_fused_csc_pass": 10.0 * Z + 2 * flags, "ccd_fused_csr_pass": 10.0 * Z + 2 * flags,
            "ccd_flat_sweep": 6.0 * Z + 2 * flags, "ccd_flat_resid": 10.0 * Z + 2 * flags,
            }
    # scatter: 16-bit local index + segment id (one-byte step + a base per tile, or 32 bits: layout kind
    # "scatter32") + fp32 value read + written.  The v-pass streams the row-major (CSR) copy, the u-pass the other.
    for kn, side, rw in (("ccd_scatter_v_pass", "csr", 8.0), ("ccd_scatter_u_pass", "csc", 8.0), ("ccd_scatter_sweep", "csr", 4.0)):
        phys[kn] = (2.0 + rw + (4.0 if layout[side]["kind"] == "scatter32" else 1.0 + 4.0 / 256)) * Z
    roofline = None
    dom = None
    if ktimes:
        cand = {kname: v for kname, v in ktimes.items() if kname in alg}
        if cand:
            dom = max(cand, key=lambda kn: cand[kn][0])
            secs, launches = cand[dom]
            # a pass launched by panel groups (sharded scatter column pass): the contract bytes are those of the whole pass
            per_pass = launches // (a.k * a.steps) if (dom == "ccd_scatter_v_pass" and launches > a.k * a.steps and launches % (a.k * a.steps) == 0) else 1
            launches //= per_pass
            avg = secs / max(1, launches)
            achieved = alg[dom] / avg / 1e9
            # HBM bytes per launch from the PMC counters: NOT measured in this run (counters need their own rocprofv3
            # passes) but read from the committed profiles/traffic.json -- and only when that file was collected for
            # this kernel at this size from these kernel sources; `traffic_source` says which it was
            traffic, traffic_source = None, "none: no PMC record for this kernel and size in profiles/traffic.json"
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    ent = tj.get(f"{dom}@{Z}") or tj.get(dom)
                    if ent and int(ent.get("nnz", -1)) == Z and int(ent.get("rows", m)) == m and int(ent.get("cols", n)) == n:
                        if ent.get("kernel_src_sha16") == kernel_source_hash():
                            traffic = ent.get("hbm_bytes_per_launch")
                            traffic_source = (f"profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                              f"{ent.get('collected', '?')}, kernel sources {ent.get('kernel_src_sha16')}")
                        else:
                            traffic_source = ("stale: the kernel sources changed since profiles/traffic.json was collected "
                                              f"({ent.get('kernel_src_sha16')} -> {kernel_source_hash()}); rerun tools/prof_final.sh")
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "traffic_source": traffic_source, "avg_launch_us": round(avg * 1e6, 2), "launches": int(launches), "launches_per_pass": int(per_pass),
                        "algorithmic_bytes_per_launch": int(alg[dom]),
                        "streamed_bytes_per_launch": int(phys[dom]) if (dom in phys and (
                            "scatter" in dom or layout["csr" if "csr" in dom else "csc"]["kind"] == "lds")) else int(alg[dom]),
                        "as_written_equiv_frac": round(a.k * (48 + 16 * a.inner) * nnz_global /
                                                       (elapsed / a.steps) / 1e9 / (HBM_PEAK_GBS * world), 4)}
    kernels = {kn: {"total_ms": round(v[0] * 1e3, 3), "launches": int(v[1]),
                    "avg_us": round(v[0] / max(1, v[1]) * 1e6, 2)} for kn, v in (ktimes or {}).items()}

    rep = solver.iterate(1, with_rmse=True)  # one more iteration, just to report a test RMSE
    rmse_now = rep[0].rmse

    # ---------------- the rank-one sweep on its own (B_r1 = 8 B/nnz per half-sweep) -------
    # In the default schedule at T = 1 the first sweep of every rank is fused into the two passes; the
    # standalone kernel only runs for inner iterations 2..T.  One outer iteration at T = 2 on a second
    # solver over the same resident inputs gives its launch time: k v-sweeps (CSC copy) + k u-sweeps (CSR).
    rank_one = None
    if world == 1 and not a.no_rank_one and not a.no_event_pass and a.schedule == 1:
        solver.close()
        p2 = mfx.parameter()
        p2.k, p2.lambda_, p2.maxinneriter, p2.device = a.k, a.lam, 2, local_rank
        p2.schedule, p2.kernel_variant, p2.tiles_per_span = a.schedule, a.variant, a.tiles
        p2.panel_rows, p2.wg_waves, p2.graph, p2.layout_build, p2.profile = a.panel_rows, a.wg_waves, -1, a.layout_build, 1
        s2 = mfx.CcdSolver(None, None, p2, device_arrays=d)
        s2.set_factors(W0)
        s2.iterate(1, with_rmse=False)
        kt2 = s2.kernel_times()
        s2.close()
        sweep_name = "ccd_flat_sweep" if "ccd_flat_sweep" in kt2 else "ccd_scatter_sweep"
        if sweep_name in kt2:
            secs, launches = kt2[sweep_name]
            avg = secs / max(1, launches)
            b_r1 = 8.0 * Z + 0.5 * ((4.0 * (n + 1) + 4.0 * m + 4.0 * n) + (4.0 * (m + 1) + 4.0 * n + 4.0 * m))  # mean of the two sides
            rank_one = {"kernel": sweep_name, "avg_launch_us": round(avg * 1e6, 2), "launches": int(launches),
                        "algorithmic_bytes_per_launch": int(b_r1), "achieved": round(b_r1 / avg / 1e9, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(b_r1 / avg / 1e9 / HBM_PEAK_GBS, 4),
                        "note": "standalone v-/u-sweep launches of one outer iteration at T = 2 (inner iteration 2)"}

    # ---------------- ALS (BASELINE configs[3]) on the same resident matrix: 3 iterations at k = 64 ----------------
    als = None
    if world == 1 and not a.no_als and a.workload == "netflix" and a.k == 64 and a.schedule == 1 and not a.no_event_pass:
        solver.close()
        m_als = als_measure(a, d, mfx, 64, 3, 1)
        als = {"ms_per_iteration": m_als["ms_per_iteration"], "k": 64, "steps": 3, "warmup": 1, "workload": m_als["workload"],
               "roofline": m_als["roofline"], "half_sweeps": m_als["half_sweeps"], "rmse": m_als["rmse"], "als_src_sha16": m_als["als_src_sha16"]}

    # ---------------- CPU baseline: the oracle on this box's host cores (rank 0, N = 1 only) --------
    cpu_baseline = None
    if world == 1 and not a.no_cpu_baseline:
        from oracle import oracle as orc
        # bounded sample: at most ~1.3e8 ratings -- beyond that the leading nnz-balanced block of user rows (the CPU
        # cost per rating does not depend on how many blocks follow), SURVEY 8d "for Z = 1e9 run the CPU on a 1/8 shard"
        cap = 130_000_000
        blocks = max(1, -(-Z // cap))
        dc = d if blocks == 1 else synth_torch.leading_row_block(d, blocks)
        host = synth_torch.to_rating_data(dc)
        del dc
        threads = orc.max_threads()  # min(OpenMP max, cgroup/affinity share of this box, 32)
        ks = max(1, min(a.cpu_ranks, a.k))
        Wc = np.ascontiguousarray(W0[:ks, :host.rows])
        _, _, _, times, _, _ = orc.ccdr1(host, Wc, ks, a.lam, 2, a.inner, threads)
        t_steady = float(times[1].sum())  # outer iteration 2: includes the add-back (src/CCD.cpp:100)
        t_outer_k = t_steady * (a.k / ks)
        cpu_baseline = {"value": round(host.nnz / t_outer_k, 1), "unit": "nnz/s", "cores": threads, "kind": "port",
                        "sample": f"{ks} of {a.k} ranks, outer iterations 1-2 on " +
                                  ("the full matrix" if blocks == 1 else f"the leading 1/{blocks} nnz-balanced block of user rows "
                                   f"({host.rows} rows, {host.nnz} ratings)") +
                                  f"; steady-state iteration 2 ({t_steady:.2f} s) scaled by {a.k}/{ks}"}
    solver.close()  # (idempotent)
    rows_per_gpu = int(d["rows"])

    # ---------------- N > 1: the strong-scaling workload of the north star on the same ranks ----------------
    strong_leg = None
    if (world > 1 or a.force_strong) and a.workload == "netflix" and not a.no_strong:
        del d, col_cnt
        torch.cuda.empty_cache()
        if world == 1:  # --force-strong: the leg's collectives through one-rank groups
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
            if comm is None:
                comm = mfx.Comm(mfx.Comm.unique_id(), 0, 1, local_rank)
        strong_leg = config5_strong_leg(a, torch, dist, mfx, synth_torch, comm, world, rank, local_rank, dev)
        if world == 1:
            comm.close()
            dist.destroy_process_group()

    if rank == 0:
        if strong:
            label = (f"synthetic (BASELINE configs[4]): ONE global {a.rows}x{a.cols}, nnz={nnz_global}, "
                     f"k={a.k}, T={a.inner}, lambda={a.lam}, nnz-balanced user-row blocks over {world} GPU(s)")
        elif (a.rows, a.cols, a.nnz) == (480189, 17770, 99072112):
            label = ("Netflix-shaped synthetic (BASELINE configs[2]): per-GPU "
                     f"{a.rows}x{a.cols}, nnz={nnz_local}, k={a.k}, T={a.inner}, lambda={a.lam}")
        elif a.workload == "nnz1e9":
            label = (f"synthetic, north star's 'k = 64, nnz = 1e9' point (ten Netflix-sized user blocks): per-GPU "
                     f"{a.rows}x{a.cols}, nnz={nnz_local}, k={a.k}, T={a.inner}, lambda={a.lam}")
        else:
            label = (f"synthetic {a.rows}x{a.cols} per GPU, nnz={nnz_local}, k={a.k}, T={a.inner}, lambda={a.lam} "
                     f"(user / item log-normal sigma {d_sigma[0]} / {d_sigma[1]})")
        out = {
            "metric": "rating-nnz/sec per CCD++ outer iter at k=%d" % a.k, "value": round(value, 1), "unit": "nnz/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 6),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": label,
                       "rows_per_gpu": rows_per_gpu, "cols": a.cols, "nnz_global": nnz_global, "k": a.k,
                       "inner_iters": a.inner, "schedule": "fused" if a.schedule == 1 else "as-written",
                       "parallelism": f"row-block shards x{world}" if world > 1 else "single GPU"},
            "roofline": roofline, "rank_one_kernel": rank_one, "cpu_baseline": cpu_baseline, "als": als, "config5_strong": strong_leg,
            "kernel_src_sha16": kernel_source_hash(), "als_src_sha16": als_source_hash(), "kernels": kernels,
            "allreduce_us_per_inner_iter": (kernels["rccl_allreduce"]["avg_us"] if "rccl_allreduce" in kernels else None),
            # host time to enqueue one outer iteration's launches (+ collectives) in the event-bracketed pass, next to the
            # GPU time of the same iteration: while it is smaller, launch cost is hidden behind the running kernels (the
            # sharded path launches eagerly -- no hipGraph replay around the collectives)
            "host_enqueue_ms_per_step": (round(kernels["host_enqueue_outer_iteration"]["avg_us"] / 1e3, 3)
                                         if "host_enqueue_outer_iteration" in kernels else None),
            "layout": layout, "test_rmse_after": round(rmse_now, 6), "gen_seconds": round(gen_s, 2), "setup_seconds": round(setup_s, 2),
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
