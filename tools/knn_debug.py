"""debug: batch k-NN covariance kernel (>= 100k points per launch -> k_knn_cov_coop<4>) against the brute-force kernel"""
import importlib, os, sys
sys.path.insert(0, ".")
import numpy as np, torch
reg = importlib.import_module("riv-slam_amd.registration")
rng = np.random.default_rng(1)
clouds = [torch.from_numpy(rng.uniform(-50, 50, size=(8192, 3)).astype(np.float32)).cuda() for _ in range(16)]
def covs(mode):
    os.environ["APDGICP_KNN_MODE"] = mode
    b = reg.BatchAPDGICP(reg.default_params(regularization=0))
    b.set_clouds(0, clouds)
    try:
        b.compute_covariances(); b.synchronize()
    except Exception as e:
        return "failed: " + str(e)[:60]
    return "ok"
print("pruned:", covs("pruned"))
