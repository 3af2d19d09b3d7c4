import sys, time
sys.path.insert(0, "/root/repo/cuda-recommender_amd")
import torch, mfx
from mfx import synth_torch
d = synth_torch.synth_ratings_device(480189, 17770, 99072112, seed=1234, device="cuda:0")
for sched in (1, 0):
    p = mfx.parameter(); p.k, p.lambda_, p.schedule = 64, 0.05, sched
    s = mfx.AlsSolver(None, None, p, device_arrays=d)
    s.set_factors(mfx.initial_col(17770, 64))
    s.iterate(1, with_rmse=False); torch.cuda.synchronize()
    t0 = time.perf_counter(); rep = s.iterate(2, with_rmse=True); torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("ALS schedule", sched, "ms per iteration", round(1e3 * el / 2, 2), "rmse", [round(r.rmse, 6) for r in rep], s.kernel_times())
    s.close()
