#!/usr/bin/env python3
"""Copies the judged summaries of tools/prof_final.sh from gpurun_out/final/<tag>/ into profiles/<round>_* and
rebuilds profiles/traffic.json: HBM bytes per launch of the dominant kernels from the FETCH_SIZE / WRITE_SIZE passes,
each entry tied to the shape it was measured at, the date and the hash of the kernel sources IT WAS MEASURED ON.

(r4) Provenance.  bench.py prints `kernel_src_sha16` / `als_src_sha16` -- computed from the sources it runs next to -- in every
JSON line; every profiled run of a workload directory (bench.log, stats.log, pmc_*.log) therefore carries the hash of the
build it measured.  The collector takes the hash FROM THOSE LOGS, never from the working tree; it REFUSES a workload whose
logs disagree with each other or with the tree (exit status 2, nothing of that workload is copied or recorded), and it never
re-stamps an existing entry: an entry whose raw counter values are unchanged is kept as it is, hash and date included.
(Round 3 stamped the tree's hash at collection time on counters that had been measured on a reverted experimental build.)
usage: collect_profiles.py r04 [--src gpurun_out/final] [--dst profiles]"""
import argparse, csv, datetime, glob, hashlib, json, os, subprocess, sys
csv.field_size_limit(1 << 30)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ap = argparse.ArgumentParser()
_ap.add_argument("tag", nargs="?", default="r04")
_ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out", "final"))
_ap.add_argument("--dst", default=os.path.join(ROOT, "profiles"))
_args = _ap.parse_args()
src, dst, tag = _args.src, _args.dst, _args.tag
os.makedirs(dst, exist_ok=True)
CCD_HASHED_SOURCES = ("ccd_kernels.hip", "ccd_scatter.hip", "flat_layout.hpp", "ccd_solver.hip", "layout_kernels.hip")  # = bench.py's


def kernel_source_hash():
    h = hashlib.sha256()
    for f in CCD_HASHED_SOURCES:
        h.update(open(os.path.join(ROOT, "cuda-recommender_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def als_source_hash():
    return hashlib.sha256(open(os.path.join(ROOT, "cuda-recommender_amd", "csrc", "als_solver.hip"), "rb").read()).hexdigest()[:16]


def run_hashes(d, key):
    """The values of `key` in the bench JSON lines of every log of run directory d: {log name: hash}."""
    out = {}
    for f in sorted(glob.glob(os.path.join(d, "*.log"))):
        for l in open(f, errors="replace"):
            if l.startswith("{") and key in l:
                try:
                    out[os.path.basename(f)] = json.loads(l).get(key)
                except Exception:
                    pass
    return out


refused = []


def provenance(d, wl, key, tree_hash):
    """The one hash all logs of d agree on, if it is also the tree's; otherwise None and a refusal on stderr."""
    hs = run_hashes(d, key)
    vals = set(hs.values())
    if not hs or None in vals:
        why = f"no {key} in its logs (a bench.py older than round 4?)"
    elif len(vals) != 1:
        why = f"its runs measured different builds: {hs}"
    elif vals != {tree_hash}:
        why = f"measured on kernel sources {vals.pop()}, the tree is {tree_hash}: re-run the profile set on this tree"
    else:
        return tree_hash
    refused.append(wl)
    print(f"[collect_profiles] REFUSED {wl}: {why}", file=sys.stderr)
    return None


def newest(path, pattern):  # gpurun merges into gpurun_out/ without deleting older runs: keep the newest file per pass
    fs = glob.glob(os.path.join(path, "**", pattern), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


def mean_counter(path, kern, ctr):
    f = newest(path, "*counter_collection.csv")
    if not f:
        return None
    s = n = 0
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            s += float(r["Counter_Value"]); n += 1
    return s / n if n else None


# bench kernel name -> substring of the device kernel's name
KERNELS = {"ccd_fused_csc_pass": "k_flat<2", "ccd_fused_csr_pass": "k_flat<3", "ccd_scatter_v_pass": "k_scatter<0", "ccd_scatter_u_pass": "k_scatter<1"}
traffic = {}
tpath = os.path.join(dst, "traffic.json")
if os.path.exists(tpath):
    try:
        traffic = {k: v for k, v in json.load(open(tpath)).items() if "@" in k}  # keep other shapes' entries
    except Exception:
        traffic = {}
for wl in sorted(os.listdir(src)) if os.path.isdir(src) else []:
    d = os.path.join(src, wl)
    if not os.path.isdir(d):
        continue
    is_als = wl.startswith("als")
    sha = provenance(d, wl, "als_src_sha16" if is_als else "kernel_src_sha16", als_source_hash() if is_als else kernel_source_hash())
    if sha is None:
        continue
    blog = os.path.join(d, "bench.log")
    lines = [l for l in open(blog) if l.startswith("{")] if os.path.exists(blog) else []
    if lines:
        open(os.path.join(dst, f"{tag}_bench_{wl}.json"), "w").write(lines[-1])
    f = newest(os.path.join(d, "stats"), "*kernel_stats.csv")
    if f:
        rd = list(csv.reader(open(f)))
        csv.writer(open(os.path.join(dst, f"{tag}_kernel_stats_{wl}.csv"), "w")).writerows([rd[0]] + [r for r in rd[1:] if "mfx" in r[0]])
    for x in sorted(glob.glob(os.path.join(d, "extra_*.txt"))):  # e.g. extra_overlap.txt -> <tag>_overlap_<workload>.txt
        open(os.path.join(dst, f"{tag}_{os.path.basename(x)[6:-4]}_{wl}.txt"), "w").write(open(x).read())
    pmc_dirs = sorted(p for p in glob.glob(os.path.join(d, "pmc_*")) if os.path.isdir(p))
    if pmc_dirs:
        with open(os.path.join(dst, f"{tag}_pmc_{wl}.txt"), "w") as out:
            for p in pmc_dirs:
                out.write(f"== rocprofv3 --pmc pass: {os.path.basename(p)[4:]}\n")
                out.write(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parse_pmc.py"), p, "re:k_flat|k_scatter|k_finalize|k_combine|k_als|k_sweep|k_resid|k_ref_"], capture_output=True, text=True).stdout)
    if not lines:
        continue
    bench = json.loads(lines[-1])
    cfg = bench.get("config", {})
    if "rows_per_gpu" not in cfg:
        continue
    # the PMC passes ran the same shape (possibly at a smaller k: traffic per launch does not depend on k)
    Z = int(cfg["nnz_global"]) if bench.get("n_gpus", 1) == 1 else None
    for name, kern in KERNELS.items():
        fetch_kb, write_kb = mean_counter(os.path.join(d, "pmc_fetch_size"), kern, "FETCH_SIZE"), mean_counter(os.path.join(d, "pmc_write_size"), kern, "WRITE_SIZE")
        if fetch_kb is None or write_kb is None or Z is None:
            continue
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the
        # bytes of a wide coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-byte stores.  For the scatter
        # passes a third of the read requests are 8- / 12-byte gathers, for which the half-count is not established:
        # the corrected figure is an UPPER bound there and the raw sum is kept beside it.
        old = traffic.get(f"{name}@{Z}")
        if old and old.get("fetch_size_kib_raw") == fetch_kb and old.get("write_size_kib_raw") == write_kb:
            continue  # the same measurement as recorded: kept as it is (never re-stamped)
        traffic[f"{name}@{Z}"] = {
            "nnz": Z, "rows": int(cfg["rows_per_gpu"]), "cols": int(cfg["cols"]), "workload": wl,
            "fetch_size_kib_raw": fetch_kb, "write_size_kib_raw": write_kb,
            "hbm_bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024),
            "hbm_bytes_per_launch_uncorrected": int((fetch_kb + write_kb) * 1024),
            "kernel_src_sha16": sha, "collected": datetime.date.today().isoformat(),
            "note": "2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), gfx950 FETCH_SIZE half-count correction; hash taken from the measured runs' own bench lines"}
# ALS: both half-sweeps of one iteration (k_als_gram* kernels; everything else in an iteration is negligible)
for wl in ("als", "als128"):
    d = os.path.join(src, wl)
    blog = os.path.join(d, "bench.log")
    if not os.path.exists(blog) or wl in refused:
        continue
    lines = [l for l in open(blog) if l.startswith("{")]
    if not lines:
        continue
    sha = json.loads(lines[-1]).get("als_src_sha16")
    w = json.loads(lines[-1])["config"]["workload"]  # "<rows>x<cols> nnz=<Z> k=<k>"
    Z, k = int(w.split("nnz=")[1].split()[0]), int(w.split("k=")[1])
    per_kernel = {}
    for ctr, sub in (("FETCH_SIZE", "pmc_fetch_size"), ("WRITE_SIZE", "pmc_write_size")):
        f = newest(os.path.join(d, sub), "*counter_collection.csv")
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            if "k_als_gram" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                per_kernel.setdefault(ctr, []).append(float(r["Counter_Value"]))
    if len(per_kernel) == 2:
        iters = max(1, len(per_kernel["FETCH_SIZE"]) // 2)  # two half-sweep launches per iteration
        fetch_kb, write_kb = sum(per_kernel["FETCH_SIZE"]) / iters, sum(per_kernel["WRITE_SIZE"]) / iters
        old = traffic.get(f"als_iteration_k{k}@{Z}")
        if old and old.get("fetch_size_kib_raw") == fetch_kb and old.get("write_size_kib_raw") == write_kb:
            continue
        traffic[f"als_iteration_k{k}@{Z}"] = {
            "nnz": Z, "k": k, "fetch_size_kib_raw": fetch_kb, "write_size_kib_raw": write_kb,
            "hbm_bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024), "hbm_bytes_per_launch_uncorrected": int((fetch_kb + write_kb) * 1024),
            "kernel_src_sha16": sha,
            "collected": datetime.date.today().isoformat(),
            "note": "sum over the two half-sweep kernels of one iteration; 2*FETCH_SIZE + WRITE_SIZE, gfx950 half-count correction (upper bound: most reads are 256-byte row gathers)"}
json.dump(traffic, open(tpath, "w"), indent=1)
print(json.dumps({k: (v["hbm_bytes_per_launch"], v["hbm_bytes_per_launch_uncorrected"], v["kernel_src_sha16"]) for k, v in traffic.items()}, indent=1))
if refused:
    sys.exit(2)
