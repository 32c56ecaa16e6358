"""Comparison of two optimiser traces (per LM trial: lambda, rho, y0, yi; per outer iteration: the pose) -- shared by the golden
generator (the two CPU restatements against each other) and the tests (oracle / GPU against the golden traces)."""
import numpy as np


def trace_close(a, b, tol_cost=1e-9, tol_pose=1e-7):
    """Two optimiser traces (dicts with lambda, rho, y0, yi, poses) against each other.  The costs y0 / yi to tol_cost relative;
    rho = (y0 - yi) / den is a DIFFERENCE of two costs, so its error is tol_cost times the cancellation y0 / |y0 - yi| (3e9 at the
    last iterations of a run that converges tightly); lambda follows rho through max(1/3, 1 - (2 rho - 1)^3).  Returns the largest
    normalised differences, asserting shapes only."""
    assert all(a[k].shape == b[k].shape for k in ("lambda", "rho", "y0", "yi", "poses")), {k: (a[k].shape, b[k].shape) for k in a}
    out = {"cost": 0.0, "rho": 0.0, "lambda": 0.0, "pose": 0.0}
    if len(a["rho"]):
        out["cost"] = float(max(np.abs(a["y0"] / b["y0"] - 1).max(), np.abs(a["yi"] / b["yi"] - 1).max())) / tol_cost
        amp = np.maximum(1.0, np.abs(b["y0"]) / np.maximum(np.abs(b["y0"] - b["yi"]), 1e-300))
        out["rho"] = float((np.abs(a["rho"] - b["rho"]) / (np.maximum(np.abs(b["rho"]), 1e-3) * amp)).max()) / tol_cost
        out["lambda"] = float((np.abs(a["lambda"] / b["lambda"] - 1) / amp).max()) / tol_cost
    if len(a["poses"]):
        out["pose"] = float(np.abs(a["poses"] - b["poses"]).max()) / (tol_pose * max(1.0, float(np.abs(b["poses"][:, :3, 3]).max())))
    return out


def golden_trace(golden, tag):
    return {k: golden[f"{tag}_trace_{k}"] for k in ("lambda", "rho", "y0", "yi", "poses")}
