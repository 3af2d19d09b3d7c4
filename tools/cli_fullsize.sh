# End-to-end run of the C++ driver (mfx_train) on a Netflix-shaped dataset directory: the reference's
# flow (load -> initial_col -> kernel_wrapper_ccdpp_NV -> calculate_rmse_directly) at full size.
cd $GRAFT_REPO_ROOT
D=/tmp/mfx_netflix_dir
python3 - <<'PY'
import os, sys, time
sys.path.insert(0, "cuda-recommender_amd")
import torch, mfx
from mfx import synth_torch
t0 = time.time()
d = synth_torch.to_rating_data(synth_torch.synth_ratings_device(480189, 17770, 99072112, seed=1234, device="cuda:0"))
mfx.dataset.write_dataset_dir("/tmp/mfx_netflix_dir", d)
print(f"dataset written in {time.time() - t0:.1f} s", flush=True)
PY
ls -la $D | head -5
for solver in "-CUDA" "-CUDA -ALS"; do
  echo "== mfx_train $solver -k 64 -t 5 -l 0.05"
  ./cuda-recommender_amd/mfx_train $solver -k 64 -t 5 -l 0.05 $D 2>&1 | grep -v "^$" | tail -12
done
# (r4) the driver's comparison flow at full size: product leg, then the reference-order leg (bit-identical to the reference's CPU solver),
# then the reference's own closing golden_compare between the two
for solver in "-CUDA -OMP" "-CUDA -OMP -ALS"; do
  echo "== mfx_train $solver -k 64 -t 3 -l 0.05"
  ./cuda-recommender_amd/mfx_train $solver -k 64 -t 3 -l 0.05 $D 2>&1 | grep -v "^$" | tail -16
done
rm -rf $D
