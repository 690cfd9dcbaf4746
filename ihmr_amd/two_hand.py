"""Two-hand MANO forward on the seam-A path: the reference's ``get_mano_output``
(``src/models/optimize_model.py:171-232``, same code in ``mlp_model.py:234-295``) expressed with the HIP MANO
layer (:mod:`ihmr_amd.mano`) plus device-side tensor glue.  Differentiable; used where the caller wants
autograd semantics (drop-in tests, data synthesis).  The refinement loop itself uses the fused kernels
(:mod:`ihmr_amd.optimize_model`)."""
import torch

TIP_IDS = (744, 320, 443, 554, 671)  # optimize_model.py:99
_CONST = {}


def _consts(device):
    """Small device constants, built once per device (no host-to-device copy per call: the forward stays capturable)."""
    if device not in _CONST:
        _CONST[device] = (torch.tensor([1.0, -1.0, -1.0], device=device), torch.tensor([-1.0, 1.0, 1.0], device=device),
                          torch.tensor(TIP_IDS, dtype=torch.long, device=device))
    return _CONST[device]


def two_hand_forward(mano_right, right_orient, left_orient, right_pose, left_pose, right_shape, left_shape, hand_trans):
    bs = right_orient.shape[0]
    sgn, flip, tips = _consts(right_orient.device)
    left_orient_f = left_orient * sgn
    left_pose_f = (left_pose.reshape(bs * 15, 3) * sgn).reshape(bs, 45)
    out = mano_right(global_orient=torch.cat([right_orient, left_orient_f], 0),
                     hand_pose=torch.cat([right_pose, left_pose_f], 0),
                     betas=torch.cat([right_shape, left_shape], 0))
    verts = out.vertices
    joints = torch.cat([out.joints, verts.index_select(1, tips)], dim=1)
    rv, rj = verts[:bs], joints[:bs]
    lv, lj = verts[bs:] * flip, joints[bs:] * flip
    shift = hand_trans.reshape(bs, 1, 3) + (rj[:, 0:1, :] - lj[:, 0:1, :])
    return rv, lv + shift, torch.cat([rj, lj + shift], dim=1)


def forward_from_packed(mano_right, pose96, shape20, trans3):
    """pose96 = [R orient 3 | R pose 45 | L orient 3 | L pose 45] (baseline_model.py:262-270 layout)."""
    return two_hand_forward(mano_right, pose96[:, 0:3], pose96[:, 48:51], pose96[:, 3:48], pose96[:, 51:96],
                            shape20[:, :10], shape20[:, 10:], trans3)
