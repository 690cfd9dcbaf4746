"""Refinement strategies (stage lists) for IHMR-OPT and IHMR-MLP.

Same structure and values as the reference's ``src/strategies/opt_default.py`` (4 stages, lines
3-78) and ``mlp_default.py`` (6 stages), registered under the same names as
``src/strategies/__init__.py:21-24``.  A stage is a dict with ``update_params``, ``loss_weights``,
``lr``, ``epoch``, ``filter_loss`` ([(loss_name, criterion)]) and ``select_loss``.

``make_opt_strategy(epoch)`` re-parameterises the per-stage iteration count: the reference default is
``epoch=300`` (4 x 301 = 1204 iterations); BASELINE.json's "200-iter" configuration is ``epoch=49``
(4 x 50 = 200 iterations).
"""
from __future__ import annotations

import copy


def _opt_weights(j2d, trans, coll, finger):
    return dict(joints_2d_loss=j2d, joints_3d_loss=1000.0, trans_loss_weight=trans,
                shape_reg_loss_weight=0.1, collision_loss_weight=coll, finger_reg_loss_weight=finger)


_OPT_FILTER = [("joints_3d_loss_p", "+0"), ("collision_loss", "-10")]


def make_opt_strategy(epoch: int = 300):
    rows = [
        (["pred_hand_trans"], _opt_weights(100.0, 1000.0, 0.1, 0.0), 1e-4),
        (["pred_left_orient", "pred_right_orient"], _opt_weights(10.0, 100.0, 1.0, 0.0), 1e-2),
        (["pred_left_pose_params", "pred_right_pose_params"], _opt_weights(10.0, 100.0, 1.0, 100000.0), 1e-2),
        (["pred_left_shape_params", "pred_right_shape_params"], _opt_weights(10.0, 100.0, 1.0, 0.0), 1e-2),
    ]
    return [dict(update_params=list(p), loss_weights=dict(w), lr=lr, epoch=int(epoch),
                 filter_loss=list(_OPT_FILTER), select_loss="joints_3d_loss_p") for p, w, lr in rows]


def _mlp_weights(j3d, trans):
    return dict(joints_2d_loss=10.0, joints_3d_loss=j3d, mano_pose_loss=10.0, mano_shape_loss=10.0,
                hand_trans_loss=trans, shape_reg_loss=0.1, shape_residual_loss=0.0, collision_loss=1.0)


def make_mlp_strategy():
    both = [("joints_3d_loss_p", "+0"), ("collision_loss", "+0")]
    rows = [
        (["pred_hand_trans"], _mlp_weights(1000.0, 1000.0), 2, both, "collision_loss"),
        (["pred_left_orient"], _mlp_weights(10.0, 10.0), 2, both, "collision_loss"),
        (["pred_right_orient"], _mlp_weights(10.0, 10.0), 2, both, "collision_loss"),
        (["pred_left_pose_params", "pred_right_pose_params"], _mlp_weights(10.0, 10.0), 2, both, "collision_loss"),
        (["pred_left_shape_params", "pred_right_shape_params"], _mlp_weights(10.0, 10.0), 2, both, "collision_loss"),
        (["pred_cam_params"], _mlp_weights(10.0, 10.0), 5, [("joints_2d_loss_p", "+0")], "joints_2d_loss_p"),
    ]
    return [dict(update_params=list(p), loss_weights=dict(w), lr=1e-4, lr_decay_type="cosine", epoch=e,
                 filter_loss=list(f), select_loss=s) for p, w, e, f, s in rows]


opt_default = make_opt_strategy(300)
mlp_default = make_mlp_strategy()
strategies = dict(opt_default=opt_default, mlp_default=mlp_default)


def get_strategy(name: str, epoch: int | None = None):
    if name in ("opt_default", "default") and epoch is not None:
        return make_opt_strategy(epoch)
    return copy.deepcopy(strategies["opt_default" if name == "default" else name])


# default (reporting) weights of the final forward, reference optimize_model.py:83-94
OPT_DEFAULT_LOSS_WEIGHTS = dict(
    joints_2d_loss=10.0, joints_3d_loss=1000.0, trans_loss_weight=100.0,
    shape_reg_loss_weight=0.1, collision_loss_weight=1.0, finger_reg_loss_weight=100000.0)
