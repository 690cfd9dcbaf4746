"""ORACLE (test infrastructure) -- numpy restatement of the reference's evaluation metrics
(``src/utils/metric_utils.py:23-38,107-143``; ``src/utils/evaluator.py:149-181``), pinned by
``tests/golden/metrics.npz``."""
import numpy as np


def single_joints_error(pred, gt, valid, scale):
    """metric_utils.py:23-38 -- MPJPE per hand; NOTE the root subtraction is cumulative on the same
    copies (first the right wrist, then -- on the already shifted arrays -- the left wrist)."""
    a, b = pred.copy(), gt.copy()
    errs = []
    for root in (0, 21):
        if valid[root, 0] > 0:
            a -= a[root:root + 1]
            b -= b[root:root + 1]
            for j in range(21):
                if valid[root + j, 0] > 0:
                    errs.append(np.linalg.norm(a[root + j] - b[root + j]) / scale)
    return errs


def pa_no_rot_inter_joints_error(pred, gt, valid, scale):
    """metric_utils.py:107-143 with use_rot=False: per-axis mean/std alignment over the valid joints."""
    v = valid[:, 0] if valid.ndim == 2 else valid
    if np.sum(v) < 2.0:
        return []
    p, g = pred[v > 0, :3], gt[v > 0, :3]
    p_al = (p - p.mean(0, keepdims=True)) / p.std(0, keepdims=True) * g.std(0, keepdims=True) + g.mean(0, keepdims=True)
    return (np.linalg.norm(p_al - g, axis=1) / scale).tolist()


def collision_stats(origin_scale, interacting):
    """evaluator.py:163-181: mean / max over the 1556 per-vertex depths x 1000 (mm), averaged over
    the interacting samples."""
    sel = origin_scale[np.asarray(interacting, bool)]
    return float(np.mean(sel.mean(1) * 1000)), float(np.mean(sel.max(1) * 1000))
