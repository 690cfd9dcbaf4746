/*
 * ihmr_hip.h -- C ABI of libihmr_hip.so, the MI355X (gfx950) implementation of the IHMR hot path.
 *
 * The reference (penincillin/IHMR) is pure Python and has NO native ABI of its own; its two native
 * seams are third-party Python modules.  Each entry point below therefore cites the reference
 * call site / Python interface it stands behind (file:line under /root/reference/src), and
 * INTEGRATION.md shows the Python-side (ctypes) binding a maintainer would add.
 *
 * Conventions:
 *   - extern "C", returns int: 0 = ok, otherwise a hipError_t value (or -1 for a bad argument);
 *     never throws.
 *   - every data pointer is a DEVICE pointer owned by the caller (PyTorch), row-major contiguous
 *     fp32 unless stated; faces / ids are int32.
 *   - asynchronous on the given hipStream_t (passed as void*); no allocation, no synchronisation,
 *     graph-capture safe.  The only allocating calls are *_create / *_destroy (model constants).
 *   - workspaces are caller-provided; their sizes come from the *_workspace_bytes queries.
 */
#ifndef IHMR_HIP_H
#define IHMR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IHMR_NUM_VERTS 778
#define IHMR_NUM_FACES 1538
#define IHMR_NUM_JOINTS 16
#define IHMR_NUM_TIPS 5
#define IHMR_SDF_GRID 32

/* ------------------------------------------------------------------ MANO model constants */
/* HOST pointers; replaces the arrays smplx 0.1.28 `MANO.__init__` registers as buffers when the
 * reference calls smplx.create(...) (models/optimize_model.py:105-106, baseline_model.py:141-142,
 * mlp_model.py:108-109). */
typedef struct ihmr_mano_arrays {
    const float* v_template;   /* (778,3) */
    const float* shapedirs;    /* (778,3,10) */
    const float* posedirs;     /* (135, 2334): row = (joint-1)*9 + 3r + c, col = 3v + k */
    const float* J_regressor;  /* (16,778) */
    const float* lbs_weights;  /* (778,16) */
    const int32_t* parents;    /* (16), parents[0] = -1 */
    const float* hands_mean;   /* (45) */
    const int32_t* faces;      /* (1538,3) */
    const int32_t* tip_ids;    /* (5): optimize_model.py:99 [744,320,443,554,671] */
} ihmr_mano_arrays;

typedef struct ihmr_mano ihmr_mano;

int ihmr_mano_create(const ihmr_mano_arrays* host, ihmr_mano** out);
int ihmr_mano_destroy(ihmr_mano* m);
/* callers mutate `.shapedirs` in place (optimize_model.py:109-113): re-upload (778,3,10) host data */
int ihmr_mano_update_shapedirs(ihmr_mano* m, const float* shapedirs_host);

/* ------------------------------------------------------------------ seam A: the MANO layer */
/* forward of `mano_model(global_orient=(N,3), hand_pose=(N,45), betas=(N,10))`
 * (optimize_model.py:194-198; smplx MANO.forward + lbs): verts (N,778,3), joints (N,16,3).
 * `workspace` (ihmr_mano_workspace_bytes(N)) keeps the skeleton records and v_posed for the backward. */
size_t ihmr_mano_workspace_bytes(int N);
int ihmr_mano_lbs_fwd(const ihmr_mano* m, const float* orient, const float* pose, const float* betas, int N,
                      float* verts, float* joints, void* workspace, void* stream);
/* backward of the above (same workspace, untouched since the forward): d_verts (N,778,3),
 * d_joints (N,16,3) -> d_orient (N,3), d_pose (N,45), d_betas (N,10).
 * need_mask bit0 = orient, bit1 = pose, bit2 = betas. */
int ihmr_mano_lbs_bwd(const ihmr_mano* m, int N, const void* workspace, const float* d_verts, const float* d_joints,
                      float* d_orient, float* d_pose, float* d_betas, int need_mask, void* stream);

/* ------------------------------------------------------------------ seam B: the collision module */
/* `SDFLoss(faces_right, faces_left, robustifier)(hand_verts (B,2,778,3), return_per_vert_loss=True,
 * return_origin_scale_loss=True)` (models/loss_utils.py:38,181-182).
 * Outputs: loss (B), per_vert (B,1556), origin_scale (B,1556), dval (B,1556,3) = d per_vert / d(query
 * vertex) (the query vertex of entry h*778+v is hand_verts[b, 1-h, v]).  robustifier <= 0 = off. */
size_t ihmr_sdf_workspace_bytes(int B);
int ihmr_sdf_collision(const int32_t* faces_right, const int32_t* faces_left, const float* hand_verts, int B,
                       float robustifier, float* loss, float* per_vert, float* origin_scale, float* dval,
                       void* workspace, void* stream);
/* The upstream module (github.com/penincillin/SDF_ihmr, unpinned commit) is absent, and the reference pins none of these
 * three conventions; all default to what DESIGN.md section 4 decides and can be switched by a maintainer who holds the real
 * package (INTEGRATION.md "Pinning seam B"):
 *   align_corners  the `align_corners` of the trilinear grid_sample of phi (0 = False, the default of torch 1.6.0)
 *   loss_divisor   loss[b] = sum of the 1556 sampled values / loss_divisor (4 = num_hands^2, the parent project's normalisation;
 *                  1 = plain sum); <= 0 = default.
 *   swap_xz        axis order of the grid as grid_sample sees it: 0 = phi[z][y][x] (a query's (x, y, z) addresses the voxel whose
 *                  centre is (x, y, z)); 1 = the grid was stored phi[x][y][z] and sampled unchanged, i.e. a query's x addresses the
 *                  field's z axis and its z the x axis.
 * (The position of the voxel centres -- cell-centred, p = -1 + (2i+1)/32 -- is NOT switchable: the exact inside / outside
 * arithmetic of the ray test is derived for it.) */
typedef struct ihmr_sdf_options { int align_corners; float loss_divisor; int swap_xz; } ihmr_sdf_options;
int ihmr_sdf_collision_ex(const int32_t* faces_right, const int32_t* faces_left, const float* hand_verts, int B,
                          float robustifier, const ihmr_sdf_options* options, float* loss, float* per_vert, float* origin_scale,
                          float* dval, void* workspace, void* stream);
/* diagnostic: dense phi grid (B,2,32,32,32) built with the product kernels (every voxel evaluated);
 * compared bit-for-bit with the oracle's grid in tests. */
int ihmr_sdf_dense_grid(const int32_t* faces_right, const int32_t* faces_left, const float* hand_verts, int B,
                        float* phi, void* workspace, void* stream);

/* ------------------------------------------------------------------ seam C: IHMR-OPT refinement */
/* Everything OptimizeModel.optimize() touches per batch (models/optimize_model.py:120-168, 235-251,
 * 390-415).  All device pointers, caller-owned.  Hand index 0 = right, 1 = left, in the reference's
 * own (un-mirrored) parametrisation. */
typedef struct ihmr_opt_io {
    /* refined parameters (updated in place) */
    float* cam;     /* (B,3)    pred_cam_params */
    float* trans;   /* (B,3)    pred_hand_trans */
    float* orient;  /* (2,B,3)  pred_{right,left}_orient */
    float* pose;    /* (2,B,45) pred_{right,left}_pose_params */
    float* shape;   /* (2,B,10) pred_{right,left}_shape_params */
    /* targets (set_input) */
    const float* init_joints_2d;    /* (B,42,3) */
    const float* init_joints_3d;    /* (B,42,4) */
    const float* init_hand_trans_j; /* (B,4) */
    const float* gt_joints_2d;      /* (B,42,3) */
    const float* gt_joints_3d;      /* (B,42,4) */
    const float* gt_hand_trans;     /* (B,4) */
    const float* hand_type_array;   /* (B,2) */
    /* outputs of forward()/__compute_loss */
    float* verts;        /* (2,B,778,3) final right / left vertices */
    float* joints_3d;    /* (B,42,3) root-aligned twice, as get_pred_result exports (:428) */
    float* joints_2d;    /* (B,42,2) projection taken before the alignment (:263) */
    float* loss_batch;   /* (8,B): 0 joints_2d_loss_p, 1 joints_3d_loss_p (both x stage weight), 2 collision
                            (masked, unweighted), 3 finger_reg, 4 gt joints_2d, 5 gt joints_3d, 6 trans_p, 7 gt trans */
    float* coll_per_vert;     /* (B,1556) */
    float* coll_origin_scale; /* (B,1556) */
    /* snapshot ring (S_max, B, *) and stage selection */
    float* snap_params;  /* (S_max, B, 122) parameter slots in IHMR_PB_* block order; only the stage's blocks are written */
    float* snap_loss;    /* (S_max, 3, B): rows IHMR_LOSS_* = joints_2d_loss_p_batch, joints_3d_loss_p_batch, collision_loss_batch */
    int32_t* selected;   /* (B) argmin index of the last stage */
    /* optimizer state, zeroed by the first kernel of every stage (a fresh torch.optim.* per stage, optimize_model.py:343-347) */
    float* adam_m;       /* (B,122) Adam exp_avg / SGD momentum buffer */
    float* adam_v;       /* (B,122) Adam exp_avg_sq */
    void* workspace;     /* ihmr_opt_workspace_bytes(B) */
    /* batch size the reference's batch-mean losses are averaged over (optimize_model.py:276-330); 0 = B.
       With norm_batch = 64 and B = k * 64 one launch carries k independent batches of 64, each with exactly the
       arithmetic of a B = 64 call (samples never interact except through these 1/batch factors). */
    int norm_batch;
    /* conventions of the collision module (ihmr_sdf_options above): 0 / 0 = defaults */
    int sdf_align_corners;
    float sdf_loss_divisor;
    int sdf_swap_xz;
    /* 1 = every iteration searches all 1538 triangles per voxel (the per-voxel candidate lists that ihmr_opt_run_stage carries from
       iteration to iteration are an exact acceleration; this switch exists to test exactly that) */
    int sdf_no_candidate_lists;
    /* 1 = a hand whose vertices cannot change during a stage (the right hand of a stage that refines only the translation) is
       re-evaluated from scratch every iteration instead of keeping what the stage's earlier iterations found out about its voxels
       (an exact acceleration as well: the same vertices give the same grid; this switch exists to test that).  Implied by
       sdf_no_candidate_lists.
       2 = only the round-5 extension is off: a left hand that a stage merely TRANSLATES (opt_default's translation stage) is otherwise
       treated as static in its own frame with a box that follows it -- the kept grid is the first iteration's, which a recomputation
       reproduces only up to the rounding of the translated vertices (~1e-7 m): within the 1e-4 parity bar, not bit-identical. */
    int sdf_no_static_reuse;
    int no_fused_tail;          /* 1: every stage runs sampling + losses, the per-hand LBS backward, the optimizer step + skeletons and
                                 * (translation / orientation stages) the skinning of the stored v_posed as separate launches -- the
                                 * checker of the fused tail launch; results are identical either way (tests/test_gpu_parity.py) */
} ihmr_opt_io;

typedef struct ihmr_opt_weights { /* strategies/opt_default.py loss_weights */
    float joints_2d, joints_3d, trans, shape_reg, collision, finger_reg;
} ihmr_opt_weights;

/* The refinable parameters of a sample are 122 slots in eight blocks, in this order (= the slot order of snap_params /
 * adam_m / adam_v); a stage's `update_params` list (strategies/opt_default.py) is a set of blocks:
 *   pred_cam_params 3 | pred_hand_trans 3 | pred_right_orient 3 | pred_left_orient 3 | pred_right_pose_params 45 |
 *   pred_left_pose_params 45 | pred_right_shape_params 10 | pred_left_shape_params 10 */
#define IHMR_OPT_NPARAM 122
enum { IHMR_PB_CAM = 1, IHMR_PB_TRANS = 2, IHMR_PB_ORIENT_R = 4, IHMR_PB_ORIENT_L = 8, IHMR_PB_POSE_R = 16, IHMR_PB_POSE_L = 32,
       IHMR_PB_SHAPE_R = 64, IHMR_PB_SHAPE_L = 128 };
/* the per-sample losses a stage may filter / select on: every non-GT loss with a `_batch` twin (utils/opt_utils.py:57-67,
 * optimize_model.py:366-371); the values are the rows of io->loss_batch */
enum { IHMR_LOSS_JOINTS_2D_P = 0, IHMR_LOSS_JOINTS_3D_P = 1, IHMR_LOSS_COLLISION = 2 };
enum { IHMR_OPTIM_ADAM = 0, IHMR_OPTIM_SGD = 1 };

/* one entry of a strategy (strategies/opt_default.py:3-78) + the two options that shape a stage */
typedef struct ihmr_opt_stage {
    int param_mask;          /* IHMR_PB_* bits: stage['update_params'] */
    int optimizer;           /* opt.optimizer: Adam(lr, betas 0.9 / 0.999, eps 1e-8) or SGD(lr, momentum 0.9) (optimize_model.py:343-347) */
    float lr;                /* stage['lr'] */
    int n_iters;             /* stage['epoch'] + 1 (optimize_model.py:397) */
    int save_freq;           /* opt.save_mid_freq: snapshot before the step at iterations 0, f, 2f, ... (:401-403) */
    int use_filter[3];       /* per IHMR_LOSS_*: stage['filter_loss'] holds a criterion on that loss */
    float filter_factor[3];  /* float32(1 + (float(criterion) + 0.1) / 100): keep snapshots with loss <= origin * factor
                                (utils/opt_utils.py:104-114); several criteria on one loss = the smallest factor */
    int select_loss;         /* IHMR_LOSS_*: stage['select_loss'], per-sample first argmin over the kept snapshots */
    int keep_lists;          /* 0: the stage rebuilds the collision kernels' candidate lists in its first iteration (self-contained whatever
                                io->workspace holds).  Non-zero = the caller's guarantee that the previous call on this io->workspace was an
                                ihmr_opt_run_stage (or its graph) of the SAME batch and B and that nothing has written the workspace since:
                                a hand then keeps its lists while its displacement test passes.  Same bits either way (the lists are an
                                exact acceleration); ~20 % fewer rebuilt voxels over opt_default */
} ihmr_opt_stage;

size_t ihmr_opt_workspace_bytes(int B);
/* `m` = right-hand model (used for both hands, optimize_model.py:194); `m_left` supplies only the left
 * faces for the collision term (loss_utils.py:34-38), NULL = use the right faces.
 * one stage of optimize() (:393-407): `n_iters` iterations of forward -> losses -> snapshot -> backward -> optimizer
 * step on the stage's parameter blocks.  Then (:377-387) filter + per-sample argmin of the select loss over the
 * snapshots, and write the selected parameters back. */
int ihmr_opt_run_stage(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B,
                       const ihmr_opt_weights* w, const ihmr_opt_stage* stage, void* stream);
/* forward() + __compute_loss(weights) only (:413-414), no step */
int ihmr_opt_forward_losses(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B,
                            const ihmr_opt_weights* w, void* stream);

/* hipGraph form of the two calls above: capture once per model instance and stage (the io pointers and the
 * per-iteration Adam constants are baked into the nodes), replay with ihmr_graph_launch.  *_create allocate
 * (graph instantiation) and must not be called inside a capture; launch is asynchronous on `stream`. */
typedef struct ihmr_graph ihmr_graph;
int ihmr_opt_stage_graph_create(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B,
                                const ihmr_opt_weights* w, const ihmr_opt_stage* stage, ihmr_graph** out);
int ihmr_opt_forward_graph_create(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B,
                                  const ihmr_opt_weights* w, ihmr_graph** out);
int ihmr_graph_launch(ihmr_graph* g, void* stream);
int ihmr_graph_destroy(ihmr_graph* g);

/* Scatter the reference's packed prediction vector `final_params` (B,122) = [cam 3 | right orient 3 | right pose 45 |
 * left orient 3 | left pose 45 | right shape 10 | left shape 10 | trans 3] (baseline_model.py:262-270,
 * mlp_model.py:426-439) into io->cam / orient / pose / shape / trans. */
int ihmr_opt_set_params(const ihmr_opt_io* io, const float* final_params, int B, void* stream);

/* diagnostics (synchronises): forward + losses once with the SDF work counters on; out4 (host) =
 * {(voxel,triangle) ray tests, exact point-triangle distances, inside voxels, needed voxels} of one
 * sdf_prep_kernel + sdf_dist_kernel launch pair -- the algorithmic work bench.py prices the roofline with. */
int ihmr_opt_sdf_stats(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B,
                       const ihmr_opt_weights* w, unsigned long long* out4, void* stream);

/* diagnostics of the fused loop.  enable = 1: zero the SDF work counters and switch them on for every launch issued or
 * recorded afterwards (a stage graph captured while they are on keeps counting); enable = 0: synchronise, copy the sixteen counter
 * slots to out16 (host) and switch them off: {ray tests, exact distances, inside voxels, needed voxels, bounding-sphere tests,
 * voxels answered from their candidate lists, voxels of such hands handed to the full search, voxels whose lists were rebuilt,
 * plane + circle tests, voxels searched in full whose candidate list was refused (longer than the list capacity, or no slot left),
 * 6 unused}. */
int ihmr_opt_sdf_counters(const ihmr_opt_io* io, int B, unsigned long long* out16, int enable);

/* diagnostics of the fused loop (synchronises): the inside-voxel bitmaps the collision kernels of the LAST launch left in the
 * workspace -- out (host) [2][B][1024] words, hand-major (right hands first), bit i of word (k * 32 + j): voxel (k, j, i) of that
 * hand's grid is read by its sample AND lies inside the mesh -- and the hands' boxes, box (host) [2][B][4] = centre xyz, scale.
 * What tests/test_gpu_parity.py counts when it puts numbers on the kept grid of a translated hand: voxels whose
 * inside / outside status differs between the kept grid and a from-scratch evaluation.  (No counterpart in the reference.) */
int ihmr_opt_sdf_inside_bits(const ihmr_opt_io* io, int B, unsigned* out, float* box);

/* ------------------------------------------------------------------ image encoder (ResNet-50 + heads) */
/* One Conv2d / Linear of `InterHandEncoder.forward` (models/networks.py:66-80, models/resnet.py:138-156) as an
 * implicit GEMM on the fp32 matrix cores: y[M = N*Ho*Wo][Cout] = act(A(x) . w + bias (+ residual)).
 * x: NHWC, pixel stride ldx floats;  w: [ceil16(kh*kw*Cin)][ldw] K-major with BatchNorm folded in, zero padded,
 * ldw a multiple of 64 (128 to use the wide tile) and >= Cout;  bias [Cout];  residual optional [M][ldr];
 * y [M][ldy];  act: 0 none, 1 ReLU, 2 sigmoid.  A Linear layer is the case H = W = kh = kw = 1.
 * Gather forms (chosen by shape): Cin % 16 == 0, ldx % 4 == 0, Cin <= 2048: 16-byte loads, one filter tap per K chunk; Cin == 4 (an
 * image padded from 3 channels), kw >= 4: 16-byte loads, one pixel of one tap per load; anything else: a scalar gather.
 * workspace (optional, device, workspace_bytes): scratch for partial sums.  Layers too small to fill the GPU (Linear layers: up to 32)
 * split their K loop over up to min(32, workspace_bytes / (M*Cout*4)) workgroups.  Layers with >= 64 K steps of 16 and 64..768 tiles
 * of 128 x 128 (at batch 64: every 3 x 3 layer and the long 1 x 1 layers from 28 x 28 down) run in Stream-K form when
 * workspace_bytes >= workers x 128 KiB (two 64 KB tile slots per worker; workers = two per CU of the device, a multiple of 8, at most
 * 512: 64 MiB on an MI355X -- exactly what the launcher checks): the workers share tiles x K steps evenly and a fix-up launch adds
 * a tile's pieces in ascending K order.  A caller with a smaller workspace gets the split-K form (different last bits, see below).  Every form sums in a fixed order: results are bit-identical from run to run; they differ between
 * forms (i.e. with and without a workspace) in the last bits -- and, because the worker count follows the device's CU count and fixes
 * the K partition, between device MODELS: bit-stable per model, not across them. */
int ihmr_conv_igemm(const float* x, const float* w, const float* bias, const float* residual, float* y, int N, int H, int W,
                    int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int ldx, int ldw, int ldy, int ldr,
                    int act, void* workspace, size_t workspace_bytes, void* stream);
/* nn.MaxPool2d(3, stride 2, padding 1) (resnet.py:107) and AvgPool2d(7) + ReLU (resnet.py:111,149-151), NHWC */
int ihmr_maxpool3x3s2(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, void* stream);
int ihmr_avgpool_relu(const float* x, float* y, int N, int HW, int C, int ldy, void* stream);

/* ------------------------------------------------------------------ evaluation metrics */
/* Per-sample partial results of the four metrics `optimize.py:98-102` prints (utils/metric_utils.py:23-38,107-143,
 * utils/evaluator.py:149-181), computed on the device from what `get_pred_result()` would export:
 * pred_joints_3d (B,42,3), gt_joints_3d (B,42,4) [xyz, weight], coll_origin_scale (B,1556) [m];
 * sample_scale (B) or NULL (= 1), interacting (B) bytes or NULL (= all interacting).
 * out6 (B,6) float64: [sum MPJPE errors, count, sum aligned errors, count, mean depth mm, max depth mm]. */
int ihmr_eval_metrics(const float* pred_joints_3d, const float* gt_joints_3d, const float* coll_origin_scale,
                      const float* sample_scale, const unsigned char* interacting, int B, double* out6, void* stream);

/* MPVPE (named by BASELINE.json; the reference exports predicted and GT meshes -- models/baseline_model.py:365-368,
 * models/mlp_model.py:708-711 -- but computes no vertex metric).  Same convention as the MPJPE above
 * (utils/metric_utils.py:23-38): per hand, root-relative, L2 per point, / scale; a mesh's root is its wrist regressed with
 * row 0 of the MANO joint regressor: root_weights (2,778) [right, left].  Meshes (B,778,3); mano_params_weight (B,2): a hand
 * counts when its weight is > 0 (a GT mesh exists).  out4 (B,4) float64: [sum right, count right, sum left, count left]. */
int ihmr_eval_mpvpe(const float* pred_right, const float* pred_left, const float* gt_right, const float* gt_left,
                    const float* root_weights, const float* mano_params_weight, const float* sample_scale, int B,
                    double* out4, void* stream);

/* ------------------------------------------------------------------ packed transfers (round 5)
 * MLPModel.set_input (models/mlp_model.py:120-170) and get_pred_result (:702-719) move 17 + 13 small tensors one copy at a time; one
 * launch does a whole table of them.  A segment = a 2-D strided copy of 32-bit words: dst[r * dst_ld + c] = src[r * src_ld + c] for
 * r < rows, c < width (all four in dwords; a contiguous tensor: rows = 1, width = its dword count; an int64 tensor counts two dwords
 * per element).  n <= IHMR_COPY_MAX_SEGS, device pointers, 4-byte aligned, no overlap between a segment's source and any destination. */
#define IHMR_COPY_MAX_SEGS 32
typedef struct ihmr_copy_seg { const void* src; void* dst; int rows, width, src_ld, dst_ld; } ihmr_copy_seg;
int ihmr_copy_segments(const ihmr_copy_seg* segs, int n, void* stream);
/* The export's GT joints (mlp_model.py:530-531: the annotation, root-aligned in place by the 3-D loss, loss_utils.py:90-98):
 * joints4 (B,42,4) [x,y,z,weight] -> out4 (B,42,4), every joint minus its sample's root joint (joint 0 if its weight > 0.5, joint 21
 * if < 1e-7, none otherwise). */
int ihmr_root_align_joints(const float* joints4, float* out4, int B, void* stream);

/* ------------------------------------------------------------------ IHMR-MLP training step */
/* Gradient of the training objective `MLPModel.compute_loss(stage['loss_weights'])` (models/mlp_model.py:514-583)
 * w.r.t. the packed prediction vector final_params (B,122) [cam 3 | R orient 3 | R pose 45 | L orient 3 | L pose 45 |
 * R shape 10 | L shape 10 | trans 3]: what `self.loss.backward()` (:586-589) delivers to the sub-network's output.
 * The state must have been set with ihmr_opt_set_params; `io->init_joints_2d / init_joints_3d` must point at the
 * ANNOTATED joints (the training terms compare with the annotation, :518-531) and io->gt_hand_trans at hand_trans (B,4).
 * One launch sequence: fused two-hand forward + 2-D / 3-D joint and collision terms, LBS backward for all parameter
 * groups, then the direct terms (_mano_pose_loss with the reference's own batch_rodrigues, _mano_shape_loss,
 * _hand_trans_loss, _shape_reg_loss, _shape_residual_loss; models/loss_utils.py:46-78,114-135) and the gather.
 * `w`: joints_2d / joints_3d / collision weights of the stage (trans, shape_reg, finger_reg must be 0 here);
 * gt_pose (B,96), gt_shape (B,20), params_weight (B,2) = mano_params_weight, init_shape (B,20),
 * trans_weight_mean (1) = mean of hand_trans[:, 0, 3] over the batch, or NULL = computed from io->gt_hand_trans (the
 * reference multiplies a (B,3) difference by a (B,1,1) weight, which broadcasts to (B,B,3): the term is
 * mean(w) * mean(d^2), mlp_model.py:557-558).
 * out_cols (n_out) int32 / d_out (B, ld_out), optional: the columns of final_params the stage's sub-network produces
 * (`update_params` in order, mlp_model.py:459-472) are also written as d_out[b][c] = grad122[b][out_cols[c]] -- the
 * dY operand of the head's backward pass.
 * Outputs: grad122 (B,122);  terms5 (B,5) per-sample shares of [mano_pose, mano_shape, hand_trans, shape_reg,
 * shape_residual] (weighted; summed over the batch they are the reference's scalars); the joint / collision terms are
 * in io->loss_batch rows 0, 1, 2 as in ihmr_opt_forward_losses. */
typedef struct ihmr_train_weights { float joints_2d, mano_pose, mano_shape, hand_trans, shape_reg, shape_residual; } ihmr_train_weights;
int ihmr_mlp_train_grad(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B,
                        const ihmr_opt_weights* w, const ihmr_train_weights* tw, const float* gt_pose, const float* gt_shape,
                        const float* params_weight, const float* init_shape, const float* trans_weight_mean,
                        float* grad122, float* terms5, const int32_t* out_cols, int n_out, float* d_out, int ld_out, void* stream);
/* Dense helpers for the backward pass of `InterHandSubNetwork` (models/networks.py:83-105; Linear-ReLU x3 + Linear):
 * the GEMMs themselves (dX = dY . W, dW = X^T . dY) run through ihmr_conv_igemm.
 *   ihmr_transpose:     y[c][r] = x[r][c]                       (rows x cols, row strides ldx / ldy)
 *   ihmr_relu_backward: dx[r][c] = y[r][c] > 0 ? dx[r][c] : 0  (nn.ReLU backward from the layer OUTPUT y)
 *   ihmr_colsum:        out[c] = sum_r x[r][c], rows in order   (bias gradient)
 *   ihmr_adam_step:     torch.optim.Adam(lr, betas, eps) step number `step` (1-based) on a flat buffer, gradient
 *                       multiplied by grad_scale first (1 / world size after a SUM all-reduce) */
int ihmr_transpose(const float* x, float* y, int rows, int cols, int ldx, int ldy, void* stream);
int ihmr_relu_backward(float* dx, const float* y, int rows, int cols, int ld_dx, int ld_y, void* stream);
int ihmr_colsum(const float* x, float* out, int rows, int cols, int ldx, void* stream);
int ihmr_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float grad_scale,
                   float lr, float beta1, float beta2, float eps, int step, void* stream);

/* ------------------------------------------------------------------ encoder training kernels (groundwork) */
/* The pieces `loss.backward()` needs for `InterHandEncoder` / `ResNet` in train mode (models/networks.py:30-80,
 * models/resnet.py:58-156), NHWC activations as in ihmr_conv_igemm, every reduction in a fixed order.
 *   ihmr_bn_train_forward:  nn.BatchNorm2d in training mode on z [M = N*H*W][C]: batch mean / biased variance per
 *       channel (two passes), y = [relu](gamma * (z - mean) * invstd + beta [+ residual]); mean, var, invstd (C) are
 *       outputs (the caller keeps them for the backward pass); running_mean / running_var (C), optional, are updated in place
 *       as nn.BatchNorm2d does: (1 - momentum) * running + momentum * batch, with the unbiased variance.
 *   ihmr_bn_train_backward: g = gradient w.r.t. the unit's output -> dz, dgamma, dbeta; relu_y = the unit's output y when a
 *       ReLU follows the BatchNorm and g has NOT been masked yet (the mask y > 0 is applied on the fly), NULL otherwise.
 *   ihmr_conv_wgrad: dW [kh*kw*Cin][ldw] (the K-major layout ihmr_conv_igemm reads) = A(x)^T . dY as an implicit GEMM on
 *       the fp32 matrix cores (Cin % 4 == 0; the 3-channel image is padded to 4); workspace holds the pixel-range partial
 *       sums ([splits][K][Cout], at least K*Cout*4 bytes).
 *   The input gradient of a convolution is ihmr_conv_igemm itself on dY with the flipped, transposed filter
 *       w'[(kh-1-fh, kw-1-fw, cout)][cin], padding k-1-pad, stride 1; for the stride-2 convolutions dY is zero-inserted
 *       first (ihmr_dilate2: [N][Ho][Wo][C] -> [N][2Ho][2Wo][C]).
 *   ihmr_maxpool3x3s2_backward / ihmr_avgpool_relu_backward: nn.MaxPool2d(3, 2, 1) (gradient to the first maximum of a
 *       window, as torch) and AvgPool2d(7) + ReLU (resnet.py:107,111,149-151). */
size_t ihmr_bn_workspace_bytes(int C);
int ihmr_bn_train_forward(const float* z, long M, int C, const float* gamma, const float* beta, const float* residual, int relu,
                          float eps, float* y, float* mean, float* var, float* invstd, float* running_mean, float* running_var,
                          float momentum, void* workspace, void* stream);
int ihmr_bn_train_backward(const float* z, const float* g, long M, int C, const float* mean, const float* invstd,
                           const float* gamma, const float* relu_y, float* dz, float* dgamma, float* dbeta, void* workspace, void* stream);
int ihmr_conv_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh,
                    int kw, int stride, int pad, int ldx, int lddy, int ldw, void* workspace, size_t workspace_bytes, void* stream);
/* the flipped, transposed filter above from the forward filter w [kh*kw*Cin][ldw] -> out [kh*kw*Cout][ldo] (rows beyond and
 * columns >= Cin untouched: zero them once) */
int ihmr_pack_dgrad_weight(const float* w, float* out, int kh, int kw, int Cin, int Cout, int ldw, int ldo, void* stream);
int ihmr_dilate2(const float* dy, float* out, int N, int Ho, int Wo, int C, void* stream);
/* stride-2 3x3 input gradient without the zeros: the four parity phases of dx are stride-1 convolutions of dY with 1x1 / 1x2 /
 * 2x1 / 2x2 sub-filters (ihmr_conv_igemm); this puts them back: dx[n][2io+pi][2jo+pj] = phase[2pi+pj][n][io][jo] */
int ihmr_interleave2(const float* p00, const float* p01, const float* p10, const float* p11, float* dx, int N, int Ho, int Wo, int C,
                     void* stream);
int ihmr_maxpool3x3s2_backward(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int Ho, int Wo, void* stream);
int ihmr_avgpool_relu_backward(const float* y, const float* dy, float* dx, int N, int HW, int C, int ldy, void* stream);

/* ------------------------------------------------------------------ image preprocessing */
/* What the reference's DataLoader workers do per image on the CPU before the encoder sees it, for a whole batch:
 * `DataProcessor.padding_and_resize` (data/data_preprocess.py:45-60: longer side -> final_size with `cv2.resize`
 * [INTER_LINEAR on uint8 BGR], zero padding bottom / right), the test-time flip of left-only samples
 * (`random_flip(do_flip=True)`, data_preprocess.py:63-72, baseline_dataset.py:71-74: image mirrored, the two hands'
 * 2-D joints swapped and x -> final_size - x), `normalize_joints_2d` (data_preprocess.py:162-169) and
 * `ToTensor` + `Normalize(0.5, 0.5)` (baseline_dataset.py:41-44,202).
 * pixels: the B raw images back to back, each (H,W,3) uint8 as `cv2.imread` returns it (BGR, row-major);
 * offsets (B) int64: byte offset of image b in `pixels`;  sizes (B,2) int32: H, W;  do_flip (B) bytes or NULL.
 * img_out (B,3,final_size,final_size) float32 in [-1,1];  img_u8 optional (B,final_size,final_size,3) = the padded
 * uint8 image itself;  joints_in / joints_out optional (B,42,3) [x, y in source pixels, weight] -> [x, y in [-1,1], weight].
 * Every image must keep at least one pixel per side after the resize (the host wrapper checks, as cv2 would raise);
 * final_size % 4 == 0. */
int ihmr_preprocess_images(const uint8_t* pixels, const int64_t* offsets, const int32_t* sizes, const uint8_t* do_flip,
                           int B, int final_size, float* img_out, uint8_t* img_u8, const float* joints_in,
                           float* joints_out, void* stream);

/* per-kernel timing hook for bench.py -- the ONE piece of mutable PROCESS-GLOBAL state of this library (everything else is stateless
 * apart from the model handle and a per-device cache of the CU count, relaxed atomics: concurrent first calls store the same value): the timer pointer and the pending event pairs are shared by every stream and every thread of the process
 * (a mutex makes concurrent callers safe, it does not separate their measurements); callers that do not set a timer never touch it.
 * When non-NULL, the library records hipEvents on the launch stream around the three
 * large kernels of a refinement iteration and accumulates here (host pointer, read after ihmr_flush_kernel_timer()):
 * slot IHMR_TIMED_SDF_PREP = sdf_prep_kernel, IHMR_TIMED_SDF_DIST = sdf_dist_kernel (every collision evaluation), IHMR_TIMED_OPT_TAIL =
 * the per-sample tail launch of a fused-loop iteration (opt_tail_kernel, any instantiation).  ms[k] / n[k]: summed event-to-event
 * time and number of timed launches; ms_event_pair / n_event_pair: EMPTY event pairs recorded right before each timed group, i.e.
 * what two event records cost by themselves -- (ms[k] / n[k] - ms_event_pair / n_event_pair) is the launch duration.  Event
 * records cannot sit inside a stream capture: the caller runs a graph-less pass. */
#define IHMR_TIMED_SDF_PREP 0
#define IHMR_TIMED_SDF_DIST 1
#define IHMR_TIMED_OPT_TAIL 2
#define IHMR_TIMED_KERNELS 4
typedef struct ihmr_kernel_timer { double ms[IHMR_TIMED_KERNELS]; long n[IHMR_TIMED_KERNELS]; double ms_event_pair; long n_event_pair; } ihmr_kernel_timer;
int ihmr_set_kernel_timer(ihmr_kernel_timer* t);
int ihmr_flush_kernel_timer(void);

/* ------------------------------------------------------------------ IHMR-MLP inference glue (models/mlp_model.py)
 * MLPModel.test() (:683-699) per stage: retrive_prev_prediction (:408-423) -> sub-network on [img_feat | final_params]
 * (networks.py:83-105) -> __update_params_single (:459-472) -> forward + losses -> select_better_params (:592-637) ->
 * save_pred_to_prev (:337-356).  Two entry points per stage replace the ~25 tensor operations the reference (and rounds 1-3 of
 * this build) ran around the fused forward:
 *   ihmr_mlp_stage_head      the four Linear layers (exact fp32 on the matrix cores, one launch per layer, 16 x 16 output tiles) on
 *                            [t->img_feat | t->final_params] -- the batch's rows of the "prev" tables, which the preceding
 *                            ihmr_mlp_forward_select keeps per batch row as well as by dataset index --, the residual added to the
 *                            stage's columns -> t->new_params (B,122), and that vector scattered into io's parameter buffers;
 *   ihmr_mlp_forward_select  = ihmr_opt_forward_losses, whose last launch also decides per sample: keep the update iff every filter
 *                            loss < prev * filter_factor (strictly) and the select loss <= prev; the kept / fallen-back row goes to
 *                            t->final_params and, with its three losses, to the tables; t->kept (B) gets the decision (mode 2).
 *                            mode 1: the evaluation of the backbone's prediction that opens test(): nothing to compare, everything
 *                            saved (+ img_feat -> img_feat_all, data_idxs_all); mode 0: evaluate only (the final state).  The
 *                            collision kernels carry their candidate lists from one call to the next (exact: checked per hand against
 *                            the pose the lists were built at); mode 1 starts them over.  mode 3 (round 6) = mode 2 for a stage that
 *                            moves neither finger pose nor shape while io's workspace still holds v_posed of exactly these finger poses
 *                            and shapes (the last call that skinned in full evaluated the same pose / shape columns for EVERY row; the
 *                            caller keeps that book): the skinning launch skips both blends, bit for bit the same vertices.  An
 *                            evaluation (mode 1-3) takes its penetration depths from the prep kernel's cell words and writes no
 *                            vertex gradient (test() has no backward).
 * Weights: K-major [Kpad][ldw] as ihmr_conv_igemm takes them (layer 0: Kpad = 1152 rows, zeros beyond 1146).  Loss indices: IHMR_LOSS_*
 * (0 joints_2d_loss_p, 1 joints_3d_loss_p, 2 collision_loss).  workspace: ihmr_mlp_workspace_bytes(B) (the hidden activations).
 * All pointers device pointers. */
typedef struct ihmr_mlp_net { const float* w[4]; const float* b[4]; int ldw[4]; int k_out; int col[122]; } ihmr_mlp_net;
typedef struct ihmr_mlp_tables {
    const int64_t* idx;        /* (B) dataset index of every batch row -- PRECONDITION: distinct within a batch (the reference's OptDataset gives
                                * padded copies their own positions, opt_dataset.py:38-51): two rows with one index would race on that row of the
                                * "prev" tables below (keep / reject reads then writes prev_loss / prev_final without ordering between rows) */
    uint8_t* data_idxs_all;    /* (num_data) */
    float* img_feat_all;       /* (num_data,1024) */
    float* prev_final;         /* (num_data,122) */
    float* prev_loss;          /* (num_data,3) columns IHMR_LOSS_* */
    const float* img_feat;     /* (B,1024) the batch's image features (first evaluation) */
    float* new_params;         /* (B,122) the parameters the next forward evaluates */
    float* final_params;       /* (B,122) the batch's state after the last decided stage */
    uint8_t* kept;             /* (B) the last stage's decisions */
} ihmr_mlp_tables;
typedef struct ihmr_mlp_stage { int n_filter; int filter_loss[4]; float filter_factor[4]; int select_loss; } ihmr_mlp_stage;
size_t ihmr_mlp_workspace_bytes(int B);
int ihmr_mlp_stage_head(const ihmr_mlp_net* net, const ihmr_mlp_tables* t, const ihmr_opt_io* io, int B, void* workspace, void* stream);
int ihmr_mlp_forward_select(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B, const ihmr_opt_weights* w,
                            const ihmr_mlp_tables* t, const ihmr_mlp_stage* stage, int mode, void* workspace, void* stream);
/* The evaluation + keep / reject decision of a stage that moved ONLY the camera (mlp_default's last stage: update `pred_cam_params`, filter
 * and select on joints_2d_loss_p; mlp_model.py:504-511,592-637) -- round 6.  The reference re-runs MANO and the collision term for it; no
 * vertex, 3-D joint or penetration depth depends on the camera, so this entry evaluates the joint losses on the raw joints of every
 * sample's ACCEPTED state (kept in `workspace` by the ihmr_mlp_forward_select calls of the same batch) with the camera in `io`, takes the
 * collision loss over from the "prev" table (what a re-evaluation returns bit for bit) and decides: one launch of one workgroup per
 * sample.  PRECONDITION: an ihmr_mlp_forward_select(mode 1) for this batch and workspace came first, and the stage's sub-network
 * touched only columns 0..2 of the 122-vector; the caller (ihmr_amd/mlp_model.py) checks the latter. */
int ihmr_mlp_camera_select(const ihmr_opt_io* io, int B, const ihmr_opt_weights* w, const ihmr_mlp_tables* t, const ihmr_mlp_stage* stage,
                           void* workspace, void* stream);
/* skeletons + skinning of the parameters held in io (no collision term, no losses): io->verts (2,B,778,3).  Used for the
 * annotation's meshes of the export (mlp_model.py:497-501). */
int ihmr_opt_forward_verts(const ihmr_mano* m, const ihmr_opt_io* io, int B, void* stream);

/* checker switch (tests only): force = 1 makes the finger-pose backward use the streaming form of its pose-gradient GEMM
 * (lbs_bwd2_kernel) at every launch size; 0 restores the default (the LDS-tiled form lbs_bwd2_lds_kernel from 256 hands on).  The two
 * forms produce the same bits (tests/test_gpu_parity.py::test_lbs_bwd2_forms_are_bit_identical).  Returns the previous value. */
int ihmr_debug_force_lbs_bwd2_streaming(int force);
/* checker switch (tests only): force = 1 makes every stage skin with both blends in every iteration; 0 restores the default (a stage that
 * moves the shape but not the finger pose stores the pose-blend offsets in its first iteration and reuses them after: no pose rows read).
 * The two produce the same bits (tests/test_gpu_parity.py::test_skin_keeps_pose_offsets_bit_identically).  Returns the previous value. */
int ihmr_debug_force_full_skin(int force);

const char* ihmr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* IHMR_HIP_H */
