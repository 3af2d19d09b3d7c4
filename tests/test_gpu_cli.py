"""GPU test of the mfx_train driver: the reference's command line end to end on a golden dataset."""
import os
import re
import struct
import sys
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("solver", ["ccd", "als"])
def test_mfx_train_matches_reference_log(tmp_path, solver):
    import mfx
    g, d = load_golden("small")
    mfx.dataset.write_dataset_dir(str(tmp_path / "ds"), d)
    exe = os.path.join(ROOT, "cuda-recommender_amd", "mfx_train")
    k, lam = int(g["k"][0]), float(g["lam"][0])
    tag = "ccd_T1" if solver == "ccd" else "als"
    args = [exe, "-CUDA", "-k", str(k), "-l", repr(lam), "-t", str(int(g[tag + "__maxiter"][0])), "-T", "1",
            "-save", str(tmp_path / "model.bin")] + (["-ALS"] if solver == "als" else []) + [str(tmp_path / "ds")]
    r = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rmse = np.array([float(x) for x in re.findall(r"\[-INFO-\] iteration num \d+ .*RMSE=([0-9.]+)", r.stdout)])
    assert np.all(np.abs(rmse - g[tag + "__rmse"]) < 1e-4), (rmse, g[tag + "__rmse"])
    final = float(re.search(r"Test RMSE = ([0-9.]+)\.", r.stdout).group(1))
    assert abs(final - float(g[tag + "__final_rmse"][0])) < 1e-4
    # model file = save_mat_t(W) then save_mat_t(H)
    raw = open(tmp_path / "model.bin", "rb").read()
    m, n = struct.unpack("<qq", raw[:16])
    W = np.frombuffer(raw[16:16 + 4 * m * n], np.float32).reshape(m, n)
    ref = g[tag + "__W"] if solver == "als" else g[tag + "__W"]  # als: rows x k row-major; ccd: k x rows stored transposed
    if solver == "ccd":
        assert (m, n) == (d.rows, k)
        W = W.T
    assert np.max(np.abs(W - ref)) < 5e-3 * np.max(np.abs(ref))


@pytest.mark.parametrize("solver", ["ccd", "als"])
def test_mfx_train_cuda_and_omp_legs_agree(tmp_path, solver):
    """`-CUDA -OMP` (scripts/doit.sh's invocation): the second leg runs the reference-order parity modes on the GPU
    (ccd_reforder.hip / als_exact.hip, bit-identical to the reference's CPU solver), says so in the log, does NOT
    print a time under the OMP name, and the driver's own golden_compare passes on W and H."""
    import mfx
    g, d = load_golden("small")
    mfx.dataset.write_dataset_dir(str(tmp_path / "ds"), d)
    exe = os.path.join(ROOT, "cuda-recommender_amd", "mfx_train")
    k, lam = int(g["k"][0]), float(g["lam"][0])
    args = [exe, "-CUDA", "-OMP", "-k", str(k), "-l", repr(lam), "-t", "3", "-T", "1"] + (["-ALS"] if solver == "als" else []) + [str(tmp_path / "ds")]
    r = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "[info] CUDA Training time:" in r.stdout and "[info] reference-order GPU leg training time:" in r.stdout
    assert "OMP Training time" not in r.stdout and "no CPU solver" in r.stdout
    # golden_compare's bar is the reference's: 10 % per entry (src/extras.cpp:218-238).  CCD++ meets it on every
    # entry; ALS entries close to zero may not (the product path solves where the reference inverts), so there the
    # check must pass or stay below 1 % of the entries
    bad = [float(x) for x in re.findall(r"NO PASS! \[([0-9.]+)%\]", r.stdout)]
    assert r.stdout.count("Check... PASS!") + len(bad) == 2, r.stdout
    assert (not bad) if solver == "ccd" else all(x < 1.0 for x in bad), r.stdout
    finals = [float(x) for x in re.findall(r"Test RMSE = ([0-9.]+)\.", r.stdout)]
    tag = "ccd_T1" if solver == "ccd" else "als"
    assert len(finals) == 2 and all(abs(x - float(g[tag + "__final_rmse"][0])) < (1e-4 if solver == "ccd" else 3e-4) for x in finals)
    # the second leg is the reference's arithmetic: its final RMSE is the golden one to the printed six decimals,
    # and so is every per-iteration RMSE of its log
    assert abs(finals[1] - float(g[tag + "__final_rmse"][0])) < 1.5e-6
    second_log = r.stdout[r.stdout.index("reference-order leg on the GPU"):]
    rmse2 = np.array([float(x) for x in re.findall(r"\[-INFO-\] iteration num \d+ .*RMSE=([0-9.]+)", second_log)])
    assert rmse2.size == 3 and np.all(np.abs(rmse2 - g[tag + "__rmse"][:3]) < 1.5e-6), (rmse2, g[tag + "__rmse"])


def test_text_ratings_to_mfx_train_matches_reference_log(tmp_path):
    """N1 end to end on the GPU: MovieLens-style "user item rating" text (1-based, arbitrary line order) ->
    convert_text_ratings -> the 12-file binary directory -> mfx_train -CUDA: the per-iteration RMSE of the log
    equals the reference's on the golden dataset the text was written from."""
    import mfx
    g, d = load_golden("small")
    rng = np.random.default_rng(11)
    rows_of = np.repeat(np.arange(d.rows), np.diff(d.csr_row_ptr.astype(np.int64)))
    order = rng.permutation(d.nnz)
    with open(tmp_path / "train.txt", "w") as f:
        for q in order:
            f.write("%d %d %.9g\n" % (rows_of[q] + 1, int(d.csr_col_idx[q]) + 1, float(d.csr_val[q])))
    with open(tmp_path / "test.txt", "w") as f:
        for q in range(d.nnz_test):
            f.write("%d %d %.9g\n" % (int(d.test_row[q]) + 1, int(d.test_col[q]) + 1, float(d.test_val[q])))
    # (the largest row / column id must occur for the shape to come out right: true for this fixture)
    conv = mfx.dataset.convert_text_ratings(str(tmp_path / "train.txt"), str(tmp_path / "ds"), str(tmp_path / "test.txt"))
    assert (conv.rows, conv.cols, conv.nnz) == (d.rows, d.cols, d.nnz)
    assert np.array_equal(conv.csr_col_idx, d.csr_col_idx) and np.array_equal(conv.csc_row_idx, d.csc_row_idx)
    exe = os.path.join(ROOT, "cuda-recommender_amd", "mfx_train")
    k, lam = int(g["k"][0]), float(g["lam"][0])
    r = subprocess.run([exe, "-CUDA", "-k", str(k), "-l", repr(lam), "-t", str(int(g["ccd_T1__maxiter"][0])), "-T", "1", str(tmp_path / "ds")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rmse = np.array([float(x) for x in re.findall(r"\[-INFO-\] iteration num \d+ .*RMSE=([0-9.]+)", r.stdout)])
    assert np.all(np.abs(rmse - g["ccd_T1__rmse"]) < 1e-4), (rmse, g["ccd_T1__rmse"])


def test_sweep_harness_protocol(tmp_path):
    """tools/sweep_times.py: the K x T x repeats protocol of the reference's scripts/times.sh, JSON lines out."""
    import json
    import sys
    out = tmp_path / "res.jsonl"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_times.py"), "synth:300x200x6000", "--out", str(out),
                        "--iters", "2", "--repeats", "2", "--ks", "1", "5", "--ts", "1", "3"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rows = [json.loads(l) for l in open(out)]
    assert len(rows) == 2 * 2 * 2 and all(x["status"] == 0 and len(x["rmse"]) == 2 for x in rows)
    assert {(x["k"], x["T"]) for x in rows} == {(1, 1), (1, 3), (5, 1), (5, 3)}
    # repeats of one configuration are bitwise reproducible
    a, b = [x for x in rows if (x["k"], x["T"]) == (5, 3)]
    assert a["rmse"] == b["rmse"] and a["nnz_per_s_per_iter"] > 0


def test_mfx_train_multi_shard(tmp_path):
    """mfx_train -nGPUs 3: three user-row-block shards from one process (loopback communicator on a
    1-GPU box, RCCL when three devices exist); log and model must match the reference like the 1-GPU run."""
    import mfx
    g, d = load_golden("small")
    mfx.dataset.write_dataset_dir(str(tmp_path / "ds"), d)
    exe = os.path.join(ROOT, "cuda-recommender_amd", "mfx_train")
    k, lam = int(g["k"][0]), float(g["lam"][0])
    r = subprocess.run([exe, "-CUDA", "-nGPUs", "3", "-k", str(k), "-l", repr(lam), "-t", "3", "-T", "1",
                        "-save", str(tmp_path / "model.bin"), str(tmp_path / "ds")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CCD FAILED" not in r.stderr, r.stderr
    rmse = np.array([float(x) for x in re.findall(r"\[-INFO-\] iteration num \d+ .*RMSE=([0-9.]+)", r.stdout)])
    assert rmse.shape == (3,) and np.all(np.abs(rmse - g["ccd_T1__rmse"]) < 1e-4), (rmse, r.stdout)
    raw = open(tmp_path / "model.bin", "rb").read()
    m, n = struct.unpack("<qq", raw[:16])
    W = np.frombuffer(raw[16:16 + 4 * m * n], np.float32).reshape(m, n).T
    assert np.max(np.abs(W - g["ccd_T1__W"])) < 2e-3 * np.max(np.abs(g["ccd_T1__W"]))


def test_bench_contract_line():
    """bench.py on a small shape: ONE JSON line on stdout with the contract keys, the roofline /
    rank_one_kernel / cpu_baseline objects and the layout report."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "20000", "--cols", "3000", "--nnz", "5000000",
                          "--k", "8", "--steps", "2", "--warmup", "1", "--cpu-ranks", "2"],
                         check=True, capture_output=True, text=True, timeout=600).stdout
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "nnz/s" and d["vs_baseline"] is None
    # value and ms_per_step are two roundings of one measurement (value to 0.1 nnz/s, ms_per_step to 1e-6 ms):
    # consistency only, no assertion on how long anything took
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / d["config"]["nnz_global"] - 1.0) < 1e-4
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["rank_one_kernel"]["kernel"] == "ccd_flat_sweep" and d["rank_one_kernel"]["launches"] == 16
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert d["layout"]["csc"]["kind"] in ("lds", "cache", "plain")


def test_mfx_train_ignores_dead_flags_unless_asked(tmp_path):
    """-e / -N / -p / -q are parsed by the reference and read by none of its solvers (src/pmf.h:33-36): on the same
    command line mfx_train prints the golden RMSE trace with or without them.  With -libpmf_flags 1 they take their
    LIBPMF meaning: -p 1 -q 1 prints the line of the reference's commented block (src/CCD.cpp:141-148) after every
    rank, and the last rank's value of an outer iteration is that iteration's RMSE."""
    import mfx
    g, d = load_golden("small")
    mfx.dataset.write_dataset_dir(str(tmp_path / "ds"), d)
    exe = os.path.join(ROOT, "cuda-recommender_amd", "mfx_train")
    k, lam, t = int(g["k"][0]), float(g["lam"][0]), int(g["ccd_T1__maxiter"][0])
    base = [exe, "-CUDA", "-k", str(k), "-l", repr(lam), "-t", str(t), "-T", "1", "-e", "0.5", "-N", "1", "-p", "1", "-q", "1"]
    r = subprocess.run(base + [str(tmp_path / "ds")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rmse = np.array([float(x) for x in re.findall(r"\[-INFO-\] iteration num \d+ .*RMSE=([0-9.]+)", r.stdout)])
    assert np.all(np.abs(rmse - g["ccd_T1__rmse"]) < 1e-4) and not re.search(r"^iter \d+ rank", r.stdout, re.M)
    r = subprocess.run(base[:-8] + ["-p", "1", "-q", "1", "-libpmf_flags", "1", str(tmp_path / "ds")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = re.findall(r"^iter (\d+) rank (\d+) time ([0-9.]+) rmse ([0-9.]+)", r.stdout, re.M)
    assert [(int(a), int(b)) for a, b, _, _ in lines] == [(i, j) for i in range(1, t + 1) for j in range(1, k + 1)]
    last = np.array([float(x[3]) for x in lines]).reshape(t, k)[:, -1]
    assert np.all(np.abs(last - g["ccd_T1__rmse"]) < 1e-4), (last, g["ccd_T1__rmse"])
