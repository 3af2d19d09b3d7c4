import sys
sys.path.insert(0, "/root/repo/cuda-recommender_amd")
import numpy as np, torch, mfx
rng = np.random.default_rng(1)
for k in (64, 128):
    X = rng.uniform(-1, 1, (300, k)).astype(np.float32)
    A = (X.T @ X + 0.05 * np.eye(k, dtype=np.float32)).astype(np.float32)
    for _ in range(3):
        Ai = mfx.als_inverse(A)
    print(k, float(np.abs(Ai @ A - np.eye(k)).max()))
