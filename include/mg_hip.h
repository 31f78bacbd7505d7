/* mg_hip.h -- C ABI of the MI355X (gfx950) motion-primitive back-projection /
 * GMM-scoring library (libmg_hip.so).
 *
 * The reference (dfki-asr/morphablegraphs) is 100 % Python and has no FFI; the
 * seam this library sits behind is the duck-typed attribute
 * MotionPrimitiveModelWrapper.motion_primitive
 * (reference morphablegraphs/motion_model/motion_primitive_wrapper.py:49-53,61-85)
 * and the batched scoring seam MotionPrimitiveGenerator.evaluate_samples_using_constraints
 * (reference morphablegraphs/motion_generator/motion_primitive_generator.py:230-261).
 * Each entry point below names the reference function it replaces.  The ctypes
 * stub a maintainer adds on the reference side is shown in INTEGRATION.md.
 *
 * Conventions: extern "C", opaque handles, plain pointers and sizes, int status
 * return (0 = MG_OK, negative = error; text via mg_last_error()), no exceptions
 * cross the boundary, caller-owned buffers.  One context per (process, device);
 * one handle <-> one host thread.  Pointers named *_dev are device pointers on
 * the context's device, everything else is host memory.  All device work is
 * enqueued on the context's stream; *_host convenience entry points synchronise.
 *
 * Index layout is the reference's and is exact: frames[b][f][d] row-major with
 * d < 3 root translation and 3+4j..3+4j+3 the (w,x,y,z) quaternion of joint j;
 * coefficient flat index = coeff_idx * n_dim + d.
 */
#ifndef MG_HIP_H
#define MG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mg_context mg_context;
typedef struct mg_primitive mg_primitive;
typedef struct mg_time_grid mg_time_grid;
typedef struct mg_constraint_set mg_constraint_set;

enum {
    MG_OK = 0,
    MG_ERR_INVALID_ARGUMENT = -1,
    MG_ERR_NO_DEVICE = -2,      /* no HIP device / HIP runtime error at init  */
    MG_ERR_HIP = -3,            /* a HIP call failed; see mg_last_error()     */
    MG_ERR_UNSUPPORTED = -4,    /* shape outside what the kernels support     */
    MG_ERR_NOT_POSITIVE_DEFINITE = -5,
    MG_ERR_OUT_OF_MEMORY = -6
};

/* element types of latent inputs / score outputs */
enum { MG_F32 = 0, MG_F64 = 1 };

/* which kernel back_project_frames uses */
enum {
    MG_PATH_AUTO = 0,   /* MFMA tile kernel when it fits, else direct */
    MG_PATH_MFMA = 1,   /* LDS-staged f32 MFMA contraction + spline    */
    MG_PATH_DIRECT = 2  /* one thread per output element               */
};

/* Model arrays exactly as the reference's JSON holds them
 * (reference motion_primitive.py:126-163, _init_gmm_from_json /
 * _init_spatial_parameters_from_json).  All float64, host memory. */
typedef struct mg_primitive_desc {
    int32_t n_basis;             /* n_basis_spatial                                   */
    int32_t n_dim;               /* n_dim_spatial                                     */
    int32_t n_components;        /* L = rows of eigen_vectors_spatial                 */
    int32_t n_canonical_frames;  /* n_canonical_frames                                */
    int32_t n_gmm;               /* K = len(gmm_weights); 0 = no mixture              */
    int32_t eigen_is_transposed; /* 0: JSON layout (L, NB*D); 1: (NB*D, L) as loaded  */
    const double *eigen_vectors; /* eigen_vectors_spatial                             */
    const double *mean_vector;   /* mean_spatial_vector (NB*D)                        */
    const double *translation_maxima; /* 3 values; NULL = [1,1,1] (v3 load path)      */
    const double *knots;         /* b_spline_knots_spatial (NB+4)                     */
    const double *gmm_weights;   /* (K)                                               */
    const double *gmm_means;     /* (K, Lg)                                           */
    const double *gmm_covars;    /* (K, Lg, Lg)                                       */
    /* The reference fits the mixture over the CONCATENATED (spatial, time) latents (reference
     * construction/motion_model_constructor.py:424; motion_primitive.py:229-231 splits s[:n_s] / s[n_s:]):
     * Lg = n_gmm_dims >= n_components, 0 = n_components.  Sampling, log p and its Jacobian work on all Lg columns;
     * back projection and constraint scoring read the first n_components of every row. */
    int32_t n_gmm_dims;
    /* time model (reference motion_primitive.py:164-181, _init_time_parameters_from_json); 0 components = none */
    int32_t n_time_components;   /* columns of eigen_vectors_time                      */
    int32_t n_basis_time;        /* n_basis_time                                      */
    int32_t reserved;
    const double *eigen_vectors_time;  /* (n_basis_time, n_time_components) as the JSON holds it */
    const double *mean_time_vector;    /* (n_basis_time)                               */
    const double *knots_time;          /* b_spline_knots_time (n_basis_time + 4)       */
} mg_primitive_desc;

/* One keyframe constraint, the subset of
 * MotionPrimitiveConstraints.evaluate the fused scorer covers (reference
 * morphablegraphs/constraints/motion_primitive_constraints.py:100-122):
 *  MG_CONSTRAINT_POSITION  -> GlobalTransformConstraint._point_distance
 *      (reference .../keyframe_constraints/global_transform_constraint.py:135-143),
 *      target axes set to NaN are unconstrained (the reference's None);
 *  MG_CONSTRAINT_DIRECTION_2D -> Direction2DConstraint.evaluate_motion_spline
 *      (reference .../keyframe_constraints/direction_2d_constraint.py:42-52), heading =
 *      xz of the root quaternion applied to ref_dir, error in degrees;
 *  MG_CONSTRAINT_JOINT_POSITION -> the same _point_distance for ANY joint of a skeleton (hands, feet: the
 *      keyframe position constraints of reference two_hand_constraint.py:51-93, pose_constraint.py:48-67 go
 *      through Skeleton.nodes[joint].get_global_position): forward kinematics along the joint's chain,
 *      p = t_root + sum_i R(q_j0 q_j1 .. q_j(i-1)) offset(j_i), quaternions (w,x,y,z) normalised like
 *      transformations.quaternion_matrix does; needs a set made by mg_constraint_set_create_fk.
 *      PARITY UNPINNED: the reference's FK lives in anim_utils (absent); pinned by an independent
 *      rotation-matrix oracle and known-answer poses (tests).  A non-zero ref_dir is a point given in the joint's
 *      own frame: p + R_global(joint) ref_dir, RelativeTransformConstraint's `global_matrix . offset`
 *      (reference relative_transform_constraint.py:46-50);
 *  MG_CONSTRAINT_LOOK_AT -> LookAtConstraint.evaluate_frame (reference look_at_constraint.py:55-66): the angle
 *      in RADIANS between the joint's global orientation applied to ref_dir (REFERENCE_VECTOR (0,0,1)) and the
 *      direction from the joint's position to `target`;
 *  MG_CONSTRAINT_JOINT_MIDPOINT -> the first residual of TwoHandConstraint.get_residual_vector_frame
 *      (reference two_hand_constraint.py:66-74): |target - (p(joint) + p(joint2)) / 2|, both by the same FK (the
 *      other two residuals of that constraint are plain MG_CONSTRAINT_JOINT_POSITIONs);
 *  MG_CONSTRAINT_JOINT_ORIENTATION -> GlobalTransformConstraint._quaternion_distance
 *      (reference global_transform_constraint.py:109-121): the angle in RADIANS between the joint's global
 *      orientation (its own quaternion included) applied to ref_dir and `target` = the wanted orientation applied to
 *      ref_dir (the caller rotates; ref_dir = (0,0,1) is the reference's ORIGIN); arccos of the normalised dot
 *      product (transformations.angle_between_vectors), clamped to [-1, 1]. */
enum { MG_CONSTRAINT_POSITION = 0, MG_CONSTRAINT_DIRECTION_2D = 1, MG_CONSTRAINT_JOINT_POSITION = 2,
       MG_CONSTRAINT_JOINT_MIDPOINT = 3, MG_CONSTRAINT_JOINT_ORIENTATION = 4, MG_CONSTRAINT_LOOK_AT = 5,
       MG_CONSTRAINT_POSE = 6 /* mg_keyframe_constraint.joint = index into the mg_pose_constraint array */,
       /* Not errors but VALUES of the (aligned) motion at the keyframe, reported in the residual matrix: what the next step of
        * a graph walk is aligned to (obj_global_*: reference optimization/objective_functions.py:290-380 hands each step the
        * aligned frames of the step before).  weight_factor multiplies the value (use 1).
        *   MG_CONSTRAINT_VALUE_POSITION: component target[0] (0, 1, 2) of the root position;
        *   MG_CONSTRAINT_VALUE_HEADING:  component target[0] (0 = x, 2 = z) of the unit heading = xz of the global orientation
        *                                 of `joint` applied to ref_dir (the aligning node and reference vector of mg_alignment_desc) */
       MG_CONSTRAINT_VALUE_POSITION = 7, MG_CONSTRAINT_VALUE_HEADING = 8 };
#define MG_MAX_CHAIN 32
typedef struct mg_keyframe_constraint {
    int32_t type;
    int32_t joint;              /* MG_CONSTRAINT_JOINT_*: index into the skeleton's joints; else unused */
    double canonical_keyframe;  /* t; may be fractional (graph_walk_planner.py:203) */
    double weight_factor;
    double target[3];           /* position xyz (NaN = free) or direction (x, z, unused) */
    double ref_dir[3];          /* direction / orientation: the vector that is rotated, e.g. (0,0,1) */
    int32_t joint2;             /* MG_CONSTRAINT_JOINT_MIDPOINT: the second joint; else unused */
    int32_t reserved;
} mg_keyframe_constraint;

/* Point-cloud pose constraint, PoseConstraint.evaluate_motion_spline (reference pose_constraint.py:48-67): the global
 * positions of `joints` at the keyframe form a cloud b; it is fitted to the wanted cloud a (`points`) by the optimal
 * weighted 2-D rigid transform (rotation about y, translation in x and z: the closed form of Kovar et al.'s "Motion
 * Graphs", theta = atan2(sum w (ax bz - bx az) - (Sax Sbz - Sbx Saz) / Sw, sum w (ax bx + az bz) - (Sax Sbx + Saz Sbz) / Sw),
 * S* = weighted sums) and the error is the MEAN distance between corresponding points after the fit, plus -- with
 * has_velocity -- |velocity - (p_0(t + 1) - p_0(t))| of the first joint.  PARITY UNPINNED: the fit and the distance are
 * anim_utils' align_point_clouds_2D / calculate_point_cloud_distance (absent); the fit is pinned by its optimality
 * (tests perturb it), the rest by a NumPy oracle. */
typedef struct mg_pose_constraint {
    int32_t n_points;        /* <= MG_MAX_POSE_POINTS */
    int32_t has_velocity;
    const int32_t *joints;   /* (n_points) indices into the skeleton (the reference's node_names) */
    const double *points;    /* (n_points, 3) the wanted cloud */
    const double *weights;   /* (n_points) */
    double velocity[3];
} mg_pose_constraint;
#define MG_MAX_POSE_POINTS 64

/* The part of a skeleton forward kinematics needs (anim_utils Skeleton: nodes with parent / offset, and the
 * pose-vector layout root translation [0:3] + one quaternion per animated joint). */
typedef struct mg_skeleton_desc {
    int32_t n_joints;
    int32_t reserved;
    const int32_t *parents;       /* (n_joints) parent index, -1 for the root (joint 0 must be the root) */
    const double *offsets;        /* (n_joints, 3) offset from the parent in the parent's frame (root: ignored) */
    const int32_t *quat_channel;  /* (n_joints) first pose channel of the joint's (w,x,y,z), -1 = not animated (identity) */
} mg_skeleton_desc;

/* The 2-D alignment MotionPrimitiveConstraints.evaluate applies to every candidate outside local-coordinate mode
 * (reference motion_primitive_constraints.py:110-114, objective_functions.py:35-37 ->
 * anim_utils align_quaternion_frames_automatically): the candidate's control points are rotated about the y axis
 * so that the heading of the aligning node in its FIRST control point equals the heading in the last frame of the
 * previous motion, and translated in x and z so that its first root position lands on that frame's (y untouched);
 * heading = xz of the node's global orientation applied to ref_dir.  Rotation and translation differ per candidate.
 * The scorer never transforms control points: with c = h.b and s = h x b (h = previous heading, b = candidate
 * heading, both unit) a position becomes (c x + s z + tx, y, -s x + c z + tz), a heading (c hx + s hz, -s hx + c hz).
 * PARITY UNPINNED: anim_utils is absent; pinned by a 4x4-matrix oracle that does transform the control points.
 * joint = MG_ALIGN_START_POSE is the other branch of the reference's align_quaternion_frames (objective_functions.py:38-47,
 * taken for the first primitive of a walk, when there are no previous frames but a start pose): every candidate gets the
 * SAME rotation about y, heading = (cos, sin) of the start orientation's y angle; its first root position lands on
 * (position[0], ., position[2]) -- (0, 0) is what the reference's arithmetic produces, see candidate_scoring.py -- and every
 * height is raised by position[1].  Start orientations with x or z angles are not covered. */
#define MG_ALIGN_START_POSE (-1)
typedef struct mg_alignment_desc {
    int32_t joint;        /* skeleton.aligning_root_node as an index into the skeleton (0 = root, no skeleton needed) */
    int32_t reserved;
    double position[3];   /* root position of the previous motion's last frame (x and z are used) */
    double heading[2];    /* (x, z) heading of the aligning node in that frame; normalised by the library */
    double ref_dir[3];    /* skeleton.aligning_root_dir, e.g. (0,0,1) */
} mg_alignment_desc;

/* ---- library / context ------------------------------------------------------ */
const char *mg_version(void);
/* message of the last failed call on this thread */
const char *mg_last_error(void);
const char *mg_status_string(int status);

/* stream: a hipStream_t to enqueue on (e.g. torch's current stream) or NULL for a
 * stream owned by the context. */
int mg_context_create(int device, void *stream, mg_context **out);
void mg_context_destroy(mg_context *ctx);
int mg_context_set_stream(mg_context *ctx, void *stream);
/* The persistent frames kernel normally occupies every CU (one workgroup each, all of its LDS); leave n CUs free
 * so that kernels on other streams -- RCCL's all-gather of the scores -- run beside it instead of behind it. */
int mg_context_set_reserved_cus(mg_context *ctx, int32_t n);
/* Test and tuning switches as explicit calls (the library reads no environment variable): value 0 restores the
 * default.  They select between kernels with identical results or change the launch geometry of grids planned
 * AFTER the call; tests use them to cover the fallback kernels.
 *   MG_OPT_FORCE_VALU_SCORE   1 = constraint scoring on the VALU kernel even where the MFMA kernel applies
 *   MG_OPT_FORCE_VALU_SAMPLE  1 = mixture sampling on the VALU kernel
 *   MG_OPT_RING_SLOTS         2 = two LDS ring slots in the frames kernel even where three fit
 *   MG_OPT_CHUNK_WINDOW       n = at most n basis functions per time-chunk window (4 .. 11)
 *   MG_OPT_CHUNK_SAMPLES      n = at most n time samples per chunk (1 .. 48)
 *   MG_OPT_FRAMES_KERNEL      which LDS-staged frames kernel the MFMA path launches: 0 = by batch size (chunk-stationary
 *                             from two units per workgroup on), 1 = tile-major (units of one tile's consecutive chunks,
 *                             rows shared by neighbouring chunks carried over), 2 = chunk-stationary (a workgroup keeps one
 *                             chunk's eigenvector window in registers; MG_ERR_UNSUPPORTED where the window does not fit)
 *                             -- identical results 
 *   MG_OPT_PLACED_FAST_PCT    mg_device_malloc_placed: pattern / fill ratio (in percent) up to which a candidate counts as
 *                             fast (0 = 115); tests lower it to walk through both recipes
 *   MG_OPT_OPTIONS_STEP       mg_options_step: 0 = one launch for the whole step where every option allows it, 1 = always one
 *                             chain of launches per option (identical results; tests compare the two), 2 = one launch, and
 *                             mg_options_step_device_counts always draws its counts with a kernel in front (never takes the
 *                             counts the step before drew ahead; A/B), 3 = one launch with the release / acquire form of the
 *                             kernel's publication protocol (csrc/mg_options.hip: the fence-free default leans on gfx950's
 *                             write-through behaviour; 3 is the memory model's own protocol -- identical results, a test
 *                             runs both)
 *   MG_OPT_PLAIN_MALLOC       1 = mg_device_malloc is one hipMalloc whatever the size (0: buffers of 64 MiB and more are pieces
 *                             of the context's placed output regions, see mg_device_malloc)
 *   MG_OPT_GMM_KERNEL         mg_gmm_log_prob on the matrix pipe: 0 = by batch size, 1 = one 16-candidate tile per workgroup, fragments
 *                             from L2, 2 = persistent workgroups with the mixture's fragments in LDS (from 20 480 candidates on by
 *                             default; MG_ERR_UNSUPPORTED never: a mixture that does not fit LDS falls back to 1) -- identical results
 *   MG_OPT_SCORE_KERNEL       mg_score_constraints on the matrix pipe: 0 = by batch size, 1 = a wave per 16-candidate tile, (candidate,
 *                             constraint) pairs on the lanes, 2 = a wave per 64 candidates, a lane per candidate (from 49 152
 *                             candidates on by default) -- identical results
 *   MG_OPT_PLACED_HOLD        n > 0: the placement scan holds at most n candidates at once (default: a quarter of the free memory);
 *                             tests use it to make the scan drop candidates while it runs
 *   MG_OPT_TRAJECTORY_LANES   1 = one lane per candidate in the closest-point walks of mg_score_trajectory[_points / ies] whatever the
 *                             batch, 8 / 4 = that many lanes up to 65536 candidates (default: eight lanes while at most 28672
 *                             candidates are in flight, four up to 40960, one beyond -- the same bits in every case); only with
 *                             MG_OPT_TRAJECTORY_SEARCH = 1 (the reference's search is one lane per candidate)
 *   MG_OPT_TRAJECTORY_SEARCH  the closest-point search of the trajectory constraints (mg_score_trajectory[_points / ies], the scorer's
 *                             MG_FRAME_JOINT_TRAJECTORY): 0 = the reference's -- scipy's L-BFGS-B with a forward-difference gradient
 *                             (parameterized_spline.py:303-322) restated for one bounded variable, held to vectors the reference's own
 *                             function produced (tests/golden/trajectory_closest_point.npz) --, 1 = the monotone grid walk + Newton
 *                             refinement of rounds 2-4 (the first local minimum at or after the bound: the same point where the
 *                             distance has one basin ahead of the bound, another one where it has several)
 *   MG_OPT_ROOT_MODE          how the float32 frames kernels compute the root-translation channels (the two forms differ in
 *                             the last bits; see mg_primitive_root_mode): 0, 1 = the float64 pipeline (the default: the faster
 *                             of the two on gfx950), 2 = the mean/delta split, 3 = the split where the primitive's accuracy gate
 *                             allows it */
#define MG_OPT_FORCE_VALU_SCORE 0
#define MG_OPT_FORCE_VALU_SAMPLE 1
#define MG_OPT_RING_SLOTS 2
#define MG_OPT_CHUNK_WINDOW 3
#define MG_OPT_CHUNK_SAMPLES 4
#define MG_OPT_FRAMES_KERNEL 5
#define MG_OPT_PLACED_FAST_PCT 6
#define MG_OPT_OPTIONS_STEP 7
#define MG_OPT_PLAIN_MALLOC 8
#define MG_OPT_GMM_KERNEL 9
#define MG_OPT_SCORE_KERNEL 10
#define MG_OPT_ROOT_MODE 11
#define MG_OPT_PLACED_HOLD 12
#define MG_OPT_TRAJECTORY_LANES 13
#define MG_OPT_TRAJECTORY_SEARCH 14
#define MG_OPT_COUNT 15
int mg_context_set_option(mg_context *ctx, int32_t option, int32_t value);
/* Between _begin and _end every device constant the library uploads for this context (primitives and their
 * canonical grids: a graph's whole set of motion primitives) is bump-allocated from blocks of `block_bytes`
 * (0 = 64 MiB) instead of one hipMalloc each; a block is released when the last array in it has been destroyed
 * (or with the context).  mg_context_arena_bytes reports (reserved, used). */
int mg_context_arena_begin(mg_context *ctx, int64_t block_bytes);
int mg_context_arena_end(mg_context *ctx);
int mg_context_arena_bytes(mg_context *ctx, int64_t *reserved, int64_t *used);
int mg_context_synchronize(mg_context *ctx);
/* name (256 bytes), CU count, total bytes of the context's device */
int mg_context_device_info(mg_context *ctx, char *name, int32_t *n_cu, int64_t *total_mem);

/* Device memory helpers so a host language needs no HIP binding of its own.
 *
 * Where a LARGE KERNEL OUTPUT (the (B, T, D) frames) lands decides between two speed classes of the frames kernel's store
 * stream -- thousands of concurrent ~1 KB-piece streams: 63 vs 80 us for 404 MB stand-alone, 77 vs 97 us for the frames
 * kernel; a plain fill does not see the difference.  The class is a property of the physical memory behind an allocation
 * (the same memory mapped at other virtual addresses keeps it, offsets inside an allocation do not change it, fast memory
 * comes in multi-gigabyte runs), it shows in no translation or L2 counter, and nothing a kernel can choose changes it
 * (DESIGN.md "Placement", profiles/r03_placement/).  So the library PROBES: it times the store pattern and a plain fill on a
 * candidate allocation (~1.5 ms) and keeps the first candidate in the fast class.
 *
 * mg_device_malloc: buffers below 64 MiB are one hipMalloc.  From 64 MiB on the buffer is a piece of one of the context's
 * PLACED REGIONS: the first request of a size runs the scan once, later requests (after mg_device_free, which returns a piece
 * to its region, not to the driver) reuse the region -- an adaptor that allocates its output on every call pays for placement
 * once, and the scratch blocks behind the *_host entry points are placed the same way.  The scan is bounded: at most 32
 * candidates, and never more than a quarter of the free device memory held at once (processes sharing a GPU);
 * mg_context_trim_outputs releases the regions no piece of which is in use, mg_context_output_bytes reports what is held.
 * A caller that brings memory from elsewhere (a framework's tensor) can ask mg_device_probe_placement which class it is in. */
int mg_device_malloc(mg_context *ctx, int64_t bytes, void **out_dev);
int mg_device_free(mg_context *ctx, void *ptr_dev);
int mg_context_trim_outputs(mg_context *ctx);
int mg_context_output_bytes(mg_context *ctx, int64_t *reserved, int64_t *in_use, int32_t *n_regions, int32_t *n_fast);
/* The same buffer assembled from separately created physical chunks (HIP virtual-memory API: one reserved address
 * range, hipMemCreate per chunk_bytes, mapped back to back); one of the scan's two recipes.  Freed with mg_device_free. */
int mg_device_malloc_chunked(mg_context *ctx, int64_t bytes, int64_t chunk_bytes, void **out_dev);
/* mg_device_malloc's placed path with an explicit budget and a report.  max_candidates <= 0: the default scan (16 plain
 * allocations, then -- unless max_candidates == 1 -- twelve assembled from physical chunks of 32 / 8 / 2 MiB, then plain
 * again, 32 in all, at most a quarter of the free memory held -- and, when those were all slow, on with plain allocations up to
 * 400 candidates or six tenths of the free memory held while the scan runs: fast-class memory is sparse, about one 404 MB
 * allocation in thirty, and comes in clusters of the allocation order (on one box the first was the 181st candidate);
 * ~12 ms per candidate, once per region; max_candidates > 0 is an exact budget without that); if nothing is fast the best
 * candidate of all is returned.  info (may be NULL): [0] candidates probed BY
 * THIS CALL (0: the piece came from a region the context already had), [1] pattern time / fill time of the region, [2] its
 * pattern time in us, [3] 1 if it is in the fast class.  The contents are undefined afterwards.  Freed with mg_device_free. */
int mg_device_malloc_placed(mg_context *ctx, int64_t bytes, int32_t max_candidates, void **out_dev, double *info4);
/* the probe alone, on any device buffer of at least 64 MiB: info4 as above ([0] = 1).  It OVERWRITES the buffer (the probe
 * is a store pattern and a fill): call it before the buffer holds anything. */
int mg_device_probe_placement(mg_context *ctx, void *buf_dev, int64_t bytes, double *info4);
/* What the arena already knows about memory it handed out -- no probe, nothing written: info4 = {1 if buf_dev is (inside) a piece
 * of one of the context's placed regions, the region's pattern / fill ratio, its pattern time in us, 1 if the region is in the
 * fast class}; *tbps (may be NULL) = the pattern's rate in TB/s in the scan that classified the region (0: the request was too
 * small to fill the chip).  A region's class is decided ONCE, from its scan (fast from 5.9 TB/s on; the scan stops at the first
 * candidate of 6.0 TB/s and more), and stays: this is what mg_step_plan_for and the kernels' choice go by. */
int mg_device_placement_info(mg_context *ctx, const void *buf_dev, double *info4, double *tbps);
int mg_memcpy_h2d(mg_context *ctx, void *dst_dev, const void *src, int64_t bytes);
int mg_memcpy_d2h(mg_context *ctx, void *dst, const void *src_dev, int64_t bytes);
int mg_memset(mg_context *ctx, void *dst_dev, int value, int64_t bytes);

/* Kernel timing with HIP events on the context's stream.  enabled = n > 0 brackets every
 * n-th launch of each slot with an event pair (n = 1: every launch; an event pair costs a few
 * microseconds of stream time, so a timed region samples with n ~ 8); totals are resolved on query.
 * slot: 0 = back_project_frames, 1 = gmm_log_prob, 2 = score_constraints, 3 = argmin,
 *       4 = gmm_sample, 5 = spline_evaluate, 6 = fused step, 7 = a planner step in one launch (mg_options_step),
 *       8 = mg_joint_tracks, 9 = mg_score_frame_constraints, 10 = mg_score_trajectory[_points]. */
int mg_profile_enable(mg_context *ctx, int enabled);
int mg_profile_reset(mg_context *ctx);
int mg_profile_get(mg_context *ctx, int slot, double *total_ms, int64_t *launches);
/* the individual durations (ms) behind mg_profile_get, oldest first, at most `capacity` of them (the library
 * keeps the first 65536 per slot); *n = how many were written */
int mg_profile_get_samples(mg_context *ctx, int slot, float *out_ms, int64_t capacity, int64_t *n);

/* ---- multi-GPU: one process per GPU, one context per process ----------------------------------------
 * The only exchange on the path is the all-gather of per-rank scores.  RCCL is loaded on first use
 * (dlopen of librccl.so.1 -- the copy a host framework already has in the process, if any), so single-GPU users
 * carry no dependency.  mg_dist_unique_id on rank 0, the 128 bytes travel out of band (MPI, a file, a
 * torch.distributed broadcast), then mg_dist_init on every rank; mg_dist_all_gather runs on the context's stream:
 * gathered[r * count + i] = rank r's local[i].  (bench.py keeps torch.distributed's communicator: same library.) */
#define MG_DIST_ID_BYTES 128
int mg_dist_unique_id(void *id_out);
int mg_dist_init(mg_context *ctx, int32_t rank, int32_t n_ranks, const void *id);
int mg_dist_all_gather(mg_context *ctx, const void *local_dev, void *gathered_dev, int64_t count, int dtype);
/* `bytes` bytes of buf_dev from rank `root` to every rank's buf_dev, on the context's stream: how rank 0 -- the one process
 * that runs the reference's control flow -- hands a step's constraint values, seeds and component counts to the workers */
int mg_dist_broadcast(mg_context *ctx, void *buf_dev, int64_t bytes, int32_t root);
int mg_dist_finalize(mg_context *ctx);
/* mg_dist_preflight: everything about the set-up that can fail on one rank alone (librccl and its symbols, the context's device)
 * -- call it on every rank and exchange the outcome BEFORE any rank calls mg_dist_init: ncclCommInitRank is a collective, a rank
 * that enters it alone does not return.  mg_dist_info: the communicator as RCCL reports it, out3 = {rank, ranks, device}
 * (ncclCommUserRank / ncclCommCount / ncclCommCuDevice; -1 where a query is missing), {-1, 0, -1} without a communicator. */
int mg_dist_preflight(mg_context *ctx);
int mg_dist_info(mg_context *ctx, int32_t *out3);

/* ---- primitive -------------------------------------------------------------------
 * Replaces MotionPrimitive._initialize_from_json (reference motion_primitive.py:96-163):
 * transposes/scales/packs the eigenvectors, computes precisions_cholesky_ in float64
 * on the host exactly as sklearn's _compute_precision_cholesky does
 * (reference motion_primitive.py:141-142) and uploads all constants once. */
int mg_primitive_create(mg_context *ctx, const mg_primitive_desc *desc, mg_primitive **out);
void mg_primitive_destroy(mg_primitive *prim);
/* out[8] = {n_basis, n_dim, n_components, n_canonical_frames, n_gmm, kk (MFMA k-steps),
 *           mfma_supported, chunks of the canonical grid}; mg_primitive_info2: out[4] = {n_gmm_dims, n_time_components,
 *           n_basis_time, kk of the mixture} */
int mg_primitive_info(const mg_primitive *prim, int32_t *out8);
int mg_primitive_info2(const mg_primitive *prim, int32_t *out4);
/* How the float32 frames kernels compute the root-translation channels d < 3 of this primitive (the reference computes
 * everything in float64: motion_primitive.py:236-256, motion_spline.py:71-86; the other channels are a float32 pipeline).
 *   split = 0: the float64 pipeline -- control points and spline taps as float64 fma chains, rounded to float32 once;
 *   split = 1: the mean/delta split -- the output is linear in the latent vector, frames[f][d] = M[f][d] + delta[f][d] with
 *              M = the spline of mean' alone, a constant of the time grid evaluated ONCE in float64 on the host and kept as
 *              a float32 pair (Mhi, Mlo), and delta = the spline of E'.s alone, which rides in the ordinary float32 row
 *              tiles; out = Mhi + (Mlo + delta), two float32 additions.
 * The split is an OPTION (MG_OPT_ROOT_MODE 3; it measured 82.8 us against the float64 pipeline's 78.5 us on the fused step,
 * DESIGN.md 6.4, so the default is the float64 pipeline for every primitive); mode 3 takes it when its error estimate -- (L + 8) 2^-24 max_r sqrt(sum_k E'[r][k]^2 m2[k]) over the root rows r, m2[k] the
 * second moment of latent k under the primitive's mixture (1 without a mixture): the float32 rounding a root control point
 * of typical size can collect -- is at most 5e-6, half of the 1e-5 the float32 pose values are held to; *estimate returns
 * it; *split returns what the context's current mode means for this primitive.  MG_OPT_ROOT_MODE 2 forces the split (tests). */
int mg_primitive_root_mode(const mg_primitive *prim, int32_t *split, double *estimate);
/* (K, L, L) float64, upper triangular: sklearn's precisions_cholesky_ */
int mg_primitive_get_precisions_cholesky(const mg_primitive *prim, double *out);

/* MotionPrimitive._back_transform_gamma_to_canonical_time_function for a batch (reference motion_primitive.py:289-302):
 * out[b][i] = sum_{j <= i} exp(mean_t(j) + sum_l phi_l(j) gamma[b][l]) - 1, i = 0 .. n_canonical_frames - 1, with the mean
 * and the harmonics evaluated at the canonical frames by the FITPACK splev recurrence on the time knots (float64, host,
 * once per primitive).  gamma: (B, ld), first n_time_components columns; out: (B, n_canonical_frames) float64.
 * MG_ERR_INVALID_ARGUMENT if the primitive has no time model. */
int mg_time_function_canonical(mg_primitive *prim, const void *gamma_dev, int gamma_dtype, int64_t n_samples, int64_t ld,
                               double *out_dev);
int mg_time_function_canonical_host(mg_primitive *prim, const void *gamma, int gamma_dtype, int64_t n_samples, int64_t ld,
                                    double *out);
/* The TIME-WARPED synthesis of a batch (reference motion_primitive.py:268-319 behind back_project(s, use_time_parameters=True);
 * graph_walk.py:154-176 turns every step of a finished walk into frames this way).
 * mg_time_function_sample: per candidate the spline's time function t'(t) = {0, inverse of the canonical time function t(t') at
 * numpy.linspace(1, t(F-2), num), F - 1}, num = int(round(t(F-2)) * (1 / speed)) -- the reference passes this FLOAT to linspace, a
 * TypeError on NumPy >= 1.18; int() is what older NumPy made of it.  The inverse is the interpolating cubic through (t(i), i),
 * i = 0 .. F-1, that scipy's splrep(k=3, s=0) builds: knots at every data point but the second and the second-to-last, i.e. the
 * not-a-knot cubic, computed here from its second derivatives (the same function; 1e-12 F from FITPACK's B-spline form).
 * times (n, t_cap) float64, lengths (n) int32: samples of row b = lengths[b]; a row that would need more than t_cap samples gets
 * lengths[b] = -needed and is not written.  canonical_out (may be NULL): (n, F) float64, the canonical time functions.
 * mg_back_project_frames_at: frames of every candidate at its own times, (n, t_cap, D) float64 or float32 -- the float64 arithmetic
 * of mg_back_project_frames_f64 (control points as fma chains from the mean, FITPACK basis rows, four taps), one launch for the
 * batch; lengths NULL: every row has t_cap samples.  Device pointers. */
int mg_time_function_sample(mg_primitive *prim, const void *gamma, int dtype, int64_t n, int64_t ld, double speed, double *times, int32_t *lengths,
                            int32_t t_cap, double *canonical_out);
int mg_back_project_frames_at(mg_primitive *prim, const void *latents, int dtype, int64_t n, int64_t ld, const double *times, const int32_t *lengths,
                              int32_t t_cap, void *out, int out_dtype);

/* ---- time grids ---------------------------------------------------------------------
 * A set of canonical times with its B-spline basis rows (FITPACK splev semantics,
 * ext=0 extrapolation; reference motion_spline.py:86,92) and mean frames, prepared
 * once in float64 on the host.  mg_primitive_canonical_grid = np.linspace(0, F, F)
 * (reference motion_primitive.py:233), owned by the primitive. */
int mg_time_grid_create(mg_primitive *prim, const double *times, int32_t n_times, mg_time_grid **out);
void mg_time_grid_destroy(mg_time_grid *grid);
mg_time_grid *mg_primitive_canonical_grid(mg_primitive *prim);
int mg_time_grid_size(const mg_time_grid *grid);
/* copies the grid's tables: i0 (T) int32, weights (T,4) float64, times (T) float64; any may be NULL */
int mg_time_grid_get_tables(const mg_time_grid *grid, int32_t *i0, double *weights, double *times);

/* Global positions of joints[0 .. n_out) in every row of a (n_frames, n_dim) float64 frame block on the device, by forward
 * kinematics along each joint's chain: out_dev (n_frames, n_out, 3).  What map_motions_to_euclidean_space asks of
 * skeleton.nodes[j].get_global_position(frame) once per sample, frame and joint (reference space_partitioning/features.py:
 * 133-153, under construction/cluster_tree_builder.py:266-301).  PARITY UNPINNED (anim_utils' FK), as for the constraints. */
int mg_joint_positions(mg_context *ctx, const mg_skeleton_desc *skeleton, const int32_t *joints, int32_t n_out,
                       const double *frames_dev, int64_t n_frames, int32_t n_dim, double *out_dev);

/* ---- trajectory constraints -------------------------------------------------------------------------
 * TrajectoryConstraint.evaluate_motion_spline / get_residual_vector for the ROOT joint (reference
 * constraints/spatial_constraints/trajectory_constraint.py:79-121): per time sample of `grid` (NULL = the canonical grid,
 * i.e. get_motion_vector()) the distance from the candidate's root position to the closest point of a Catmull-Rom spline
 * through `control_points` (n_points x 3; reference splines/catmull_rom_spline.py:66-168) whose parameter is at or after
 * the previous sample's, starting at min_u (= min_arc_length / full_arc_length); error = weight * average distance.
 * errors_dev (B) float64: written, or added to with accumulate != 0 (the sum MotionPrimitiveConstraints.evaluate forms);
 * residuals_dev: NULL or (B, T) float64 = weight * distance per sample.  alignment: NULL (local coordinates), the
 * previous-frame record with the ROOT as aligning node, or a start-pose record.
 * The reference searches with scipy's L-BFGS-B from the lower bound (parameterized_spline.py:303-322); so does the device: L-BFGS-B 3.0
 * restated for one bounded variable (csrc/mg_traj_device.h), PINNED by tests/golden/trajectory_closest_point.npz -- 26 chains of 156
 * frames the reference's own function produced --: |u - u_ref| <= 2e-6 of the parameter range and |d - d_ref| <= 1e-6 max(1, d_ref) per
 * frame (measured 1.7e-7 / 2e-9; the forward-difference gradient with h = 1e-8 sets what two correct implementations can agree to).
 * MG_OPT_TRAJECTORY_SEARCH = 1 selects the deterministic walk of rounds 2-4 instead (grid walk u = k / granularity, parabola, Newton
 * steps: the local minimum of the first basin at or after the bound; 25 x faster; the reference's result where the distance has one
 * basin ahead of the bound).  The spline itself is pinned by tests/golden/trajectory_spline.npz. */
typedef struct mg_trajectory mg_trajectory;
int mg_trajectory_create(mg_primitive *prim, const double *control_points, int32_t n_points, int32_t granularity, mg_trajectory **out);
void mg_trajectory_destroy(mg_trajectory *trajectory);
int mg_score_trajectory(mg_primitive *prim, const mg_trajectory *trajectory, const mg_time_grid *grid, const void *latents_dev,
                        int latent_dtype, int64_t n_samples, int64_t ld, double min_u, double weight, const mg_alignment_desc *alignment,
                        double *errors_dev, int accumulate, double *residuals_dev);
/* n such scorers of the same batch size in ONE launch -- a planner step scores every option's candidates against the option's own
 * trajectory (reference graph_walk_planner.py:184-226 with a TrajectoryConstraint in every option's constraint list): 4096
 * candidates fill a quarter of the chip, sixteen launches in a row take sixteen times one, side by side they take four.  The
 * primitives must share a context; canonical grids; no residual vectors; alignments: NULL, or n records of which any may be NULL.
 * The same results as n calls of mg_score_trajectory, which this falls back to where the eight-lane walk does not apply. */
int mg_score_trajectories(int32_t n, mg_primitive *const *prims, const mg_trajectory *const *trajectories, const void *const *latents_dev,
                          int latent_dtype, int64_t n_samples, const int64_t *ld, const double *min_u, const double *weight,
                          const mg_alignment_desc *const *alignments, double *const *errors_dev, int accumulate);

/* The same search for positions the caller supplies: points_dev (n_samples, n_times, 3) float64 -- one joint's track from
 * mg_back_project_frames_f64 + mg_joint_positions, aligned by the caller -- for TrajectoryConstraint on joints other than the
 * root (trajectory_constraint.py:95-121 over skeleton.nodes[joint].get_global_position(frame)). */
int mg_score_trajectory_points(mg_primitive *prim, const mg_trajectory *trajectory, const double *points_dev, int64_t n_samples, int32_t n_times,
                               double min_u, double weight, double *errors_dev, int accumulate, double *residuals_dev);

/* ParameterizedSpline.find_closest_point_fast (constraints/spatial_constraints/splines/parameterized_spline.py:303-322) for a batch of
 * point sequences, chained the way TrajectoryConstraint.get_residual_vector chains it (trajectory_constraint.py:103-113: every
 * frame's search is bounded below by, and started at, the parameter the previous frame's search returned; the first by min_u):
 * points_dev (n_samples, n_times, 3) float64 -> params_dev (n_samples, n_times) the spline parameter of every frame's point,
 * distances_dev (n_samples, n_times) its distance to the frame's position, evaluations_dev (n_samples, n_times) int32 the (f, g)
 * evaluations the frame's search took (scipy's nfev / 2; 0 with the monotone walk) -- any may be NULL.  The search is the reference's
 * (MG_OPT_TRAJECTORY_SEARCH 0, the default: scipy's L-BFGS-B restated) or the monotone walk (1).  Pinned by
 * tests/golden/trajectory_closest_point.npz, which the reference's own function produced. */
int mg_trajectory_closest_points(mg_primitive *prim, const mg_trajectory *trajectory, const double *points_dev, int64_t n_samples, int32_t n_times,
                                 double min_u, double *params_dev, double *distances_dev, int32_t *evaluations_dev);

/* Constraints that walk a joint through EVERY frame of a candidate (mg_frame_constraints.hip), on float64 frames and joint tracks
 * that are on the device already (mg_back_project_frames_f64, mg_joint_positions):
 *
 * mg_align_frames -- frames_dev (n_samples, n_times, n_dim) float64, in place: every candidate turned about y and moved like
 *   MotionPrimitiveConstraints.evaluate aligns it (reference constraints/motion_primitive_constraints.py:106-116 through
 *   anim_utils' align_quaternion_frames / start-pose transform), the transform derived per candidate from ITS first control point:
 *   vals_dev (n_samples, n_vals) float64 = its root position x, z [and its heading x, z]: the residuals of
 *   MG_CONSTRAINT_VALUE_POSITION (axes 0, 2) [and MG_CONSTRAINT_VALUE_HEADING (axes 0, 2)] constraints at t = 0 with weight 1
 *   (mg_score_constraint_residuals).  alignment: the previous-frame record (n_vals 4) or a start-pose record (n_vals 2).
 *
 * mg_score_frame_constraint -- one constraint for the whole batch, one lane per candidate.  tracks_dev (n_samples, n_times,
 *   n_joints, 3) float64; errors_dev (n_samples) written or, with accumulate != 0, added to; residuals_dev NULL or (n_samples,
 *   mg_frame_constraint_width()) = weight * the constraint's residual vector.  Types (reference classes under
 *   constraints/spatial_constraints/):
 *     MG_FRAME_CA_POSITION          GlobalTransformCAConstraint, keyframe_constraints/global_transform_ca_constraint.py:33-46: the
 *                                   smallest distance of the joint to `target` over the first n_frames frames (axes with axis_on 0 ignored)
 *     MG_FRAME_DISCRETE_TRAJECTORY  DiscreteTrajectoryConstraint, discrete_trajectory_constraint.py:66-90: frame i against points_dev[i]
 *                                   (axes with axis_on 0 zeroed on both sides, frames beyond n_points count 0), averaged over all frames
 *     MG_FRAME_LOCAL_TRAJECTORY     LocalTrajectoryConstraint, keyframe_constraints/local_trajectory_constraint.py:45-78: the target is
 *                                   the point of trajectories[0] at arc length start_arc + the path walked so far; squared xz distances summed
 *     MG_FRAME_TRAJECTORY_SET       TrajectorySetConstraint, trajectory_set_constraint.py:82-104: n_joints joints, one trajectory each,
 *                                   arc0 = joint_arc_lengths, has_range / range_start / range_end = each trajectory's active range
 *     MG_FRAME_JOINT_ROTATION       JointRotationConstraint, keyframe_constraints/joint_rotation_constraint.py:55-72: tracks_dev is the
 *                                   (n_samples, n_times, n_joints = n_dim) FRAME block at the constraint's frame index (n_times 1),
 *                                   quat_channel = 3 + 4 * the joint's index among the animated joints, quaternion = the wanted rotation
 *   A TrajectoryConstraint on any joint is mg_score_trajectory_points.  PARITY UNPINNED (forward kinematics and alignment are
 *   anim_utils'); the arc-length look-up is pinned by tests/golden/trajectory_spline.npz. */
#define MG_FRAME_CA_POSITION 1
#define MG_FRAME_DISCRETE_TRAJECTORY 2
#define MG_FRAME_LOCAL_TRAJECTORY 3
#define MG_FRAME_TRAJECTORY_SET 4
#define MG_FRAME_JOINT_ROTATION 5
#define MG_FRAME_JOINT_TRAJECTORY 6   /* TrajectoryConstraint on any joint, trajectory_constraint.py:79-121: per frame the distance to the closest
                                         point of trajectories[0] at or after the previous frame's parameter (start_arc = the constraint's min_u),
                                         averaged -- mg_score_trajectory_points' arithmetic, as a member of a list (mg_score_frame_constraints) */
#define MG_FRAME_MAX_JOINTS 8
typedef struct mg_frame_constraint_desc {
    int32_t type;
    int32_t n_frames;                 /* frames evaluated, 0 = all n_times (CA_POSITION, LOCAL_TRAJECTORY, TRAJECTORY_SET) */
    int32_t n_points;                 /* DISCRETE_TRAJECTORY */
    int32_t n_joints;                 /* TRAJECTORY_SET; 1 otherwise */
    double weight;
    double target[3];                 /* CA_POSITION */
    int32_t axis_on[3];               /* CA_POSITION: axes of the target that count; DISCRETE_TRAJECTORY: constrained axes */
    int32_t quat_channel;             /* JOINT_ROTATION */
    const double *points_dev;         /* DISCRETE_TRAJECTORY: (n_points, 3) float64 on the device */
    double start_arc;                 /* LOCAL_TRAJECTORY */
    const mg_trajectory *trajectories[MG_FRAME_MAX_JOINTS];
    double arc0[MG_FRAME_MAX_JOINTS], range_start[MG_FRAME_MAX_JOINTS], range_end[MG_FRAME_MAX_JOINTS];
    int32_t has_range[MG_FRAME_MAX_JOINTS];
    double quaternion[4];             /* JOINT_ROTATION: (w, x, y, z) */
} mg_frame_constraint_desc;
int mg_align_frames(mg_primitive *prim, double *frames_dev, int64_t n_samples, int32_t n_times, const double *vals_dev, int32_t n_vals,
                    const mg_alignment_desc *alignment);
int mg_frame_constraint_width(const mg_frame_constraint_desc *constraint, int32_t n_times);
int mg_score_frame_constraint(mg_primitive *prim, const mg_frame_constraint_desc *constraint, const double *tracks_dev, int64_t n_samples,
                              int32_t n_times, int32_t n_joints, double *errors_dev, int accumulate, double *residuals_dev);
/* The per-frame constraints WITHOUT frames in memory (two launches for a whole constraint list) -- what the reference does per candidate in
 * MotionPrimitiveConstraints.evaluate (constraints/motion_primitive_constraints.py:96-118): back-project, align to the previous frames
 * (:106-116), then every constraint's evaluate_motion_spline over skeleton.nodes[joint].get_global_position(frame) of every frame
 * (spatial_constraints/trajectory_constraint.py:79-121, keyframe_constraints/global_transform_ca_constraint.py:33-46, ...):
 * mg_joint_tracks: the tracks (n, T, joints, 3) float64 that mg_back_project_frames_f64 -> mg_align_frames -> mg_joint_positions give
 * through (n, T, n_dim) float64 frames -- ~98 KB of traffic per 'walk' candidate -- from ONE launch that keeps a candidate's control
 * points of the channels the joints' chains read in LDS: the same operations on the same values (bit-identical tracks), 24 bytes per
 * (candidate, time, joint) written.  A plan (mg_track_plan_create) fixes up to 4 requests = (time grid at the call, 1 .. 8 joints);
 * align_joint = the node candidates are aligned through when mg_joint_tracks gets an alignment (0: the root; -1: never aligned).
 * mg_score_frame_constraints: a LIST of constraints over those tracks in one launch per 4 constraints, errors added in list order
 * (what mg_score_frame_constraint gives called once per constraint with accumulate; a joint's trajectory constraint, MG_FRAME_JOINT_TRAJECTORY,
 * below 65 536 candidates as a launch of mg_score_trajectory_points at its place in the order); residuals_dev: NULL, or one pointer (or
 * NULL) per constraint. */
typedef struct mg_track_plan mg_track_plan;
int mg_track_plan_create(mg_primitive *prim, const mg_skeleton_desc *skeleton, int32_t n_requests, const int32_t *n_joints, const int32_t *joints,
                         int32_t align_joint, mg_track_plan **out);
void mg_track_plan_destroy(mg_track_plan *plan);
int mg_joint_tracks(mg_track_plan *plan, const void *latents, int dtype, int64_t n_samples, int64_t ld, const mg_alignment_desc *alignment,
                    const mg_time_grid *const *grids, double *const *tracks_dev);
int mg_score_frame_constraints(mg_primitive *prim, int32_t n_constraints, const mg_frame_constraint_desc *const *constraints, const double *const *tracks_dev,
                               const int32_t *n_times, const int32_t *n_joints, int64_t n_samples, double *errors_dev, int accumulate,
                               double *const *residuals_dev);

/* A planner step's per-frame constraint lists AND the options' first minima, for options whose candidates and errors are on the device
 * already (mg_options_step drew them and scored their keyframe constraints, mg_score_trajectories added their root trajectories) -- the
 * reference scores every option's whole constraint list inside one planner step (motion_generator/graph_walk_planner.py:184-226,
 * constraints/motion_primitive_constraints.py:100-122): TWO launches whatever the number of options (every option's joint tracks;
 * every option's list + its first minimum under mg_argmin_first's rule) and one read-back of the result records.
 *   plans[k]: NULL for an option without a per-frame list (its first minimum is still taken); grids, tracks_dev:
 *   [n_options][MG_TRACK_MAX_REQUESTS] (unused requests NULL; grids NULL = the canonical grid); constraints: [n_options][MG_FRAME_LIST_MAX],
 *   n_constraints[k] <= MG_FRAME_LIST_MAX of them used (no MG_FRAME_JOINT_ROTATION: it reads a frame), request_of[k][i]: the plan's request
 *   whose tracks constraint i reads; errors_dev[k] (n_samples) float64: read, added to, written back; results_dev: record k at
 *   k * result_stride = {int64 index, float64 error, float64 latent[n_gmm_dims]}; results_host: NULL or where the records are copied.
 * The additions are mg_joint_tracks + mg_score_frame_constraints' per option, in list order: the same errors, the same winners. */
#define MG_TRACK_MAX_REQUESTS 4
#define MG_FRAME_LIST_MAX 4
int mg_options_frame_lists(int32_t n_options, mg_primitive *const *prims, mg_track_plan *const *plans, const void *const *latents_dev, int latent_dtype,
                           int64_t n_samples, const int64_t *ld, const mg_alignment_desc *const *alignments, const mg_time_grid *const *grids,
                           double *const *tracks_dev, const int32_t *n_constraints, const mg_frame_constraint_desc *const *constraints,
                           const int32_t *request_of, double *const *errors_dev, void *results_dev, int64_t result_stride, void *results_host);

/* ---- hot path, device pointers ------------------------------------------------------ */

/* MotionPrimitive.back_project(s, False).get_motion_vector() for a batch
 * (reference motion_primitive.py:206-256 + motion_spline.py:71-86), or
 * MotionSpline.evaluate(t) when grid holds arbitrary times (motion_spline.py:89-92).
 * latents: (B, ld) of latent_dtype, first n_components columns used.
 * frames_dev: (B, T, D) float32, written completely.  grid NULL = canonical grid. */
int mg_back_project_frames(mg_primitive *prim, const mg_time_grid *grid,
                           const void *latents_dev, int latent_dtype, int64_t n_samples, int64_t ld,
                           float *frames_dev, int path);

/* Same in float64 throughout (all channels), frames_dev (B, T, D) float64.  Used by the
 * single-sample adaptor calls where the reference's float64 results are expected. */
int mg_back_project_frames_f64(mg_primitive *prim, const mg_time_grid *grid,
                               const void *latents_dev, int latent_dtype, int64_t n_samples, int64_t ld,
                               double *frames_dev);

/* MotionPrimitive.back_project_spatial_coeffs for a batch
 * (reference motion_primitive.py:236-256): coeffs_dev (B, NB, D), float64 when
 * out_dtype == MG_F64 else float32. */
int mg_back_project_coeffs(mg_primitive *prim, const void *latents_dev, int latent_dtype,
                           int64_t n_samples, int64_t ld, void *coeffs_dev, int out_dtype);

/* MotionSpline.get_motion_vector()/evaluate(t) from explicit coefficient arrays
 * (reference motion_spline.py:71-92; callers overwrite spline.coeffs with aligned
 * coefficients, motion_primitive_constraints.py:113): coeffs_dev (n, NB, D) float64 ->
 * frames_dev (n, T, D) float64. */
int mg_spline_evaluate(mg_primitive *prim, const mg_time_grid *grid, const double *coeffs_dev,
                       int64_t n_splines, double *frames_dev);

/* GaussianMixture.score_samples (reference motion_primitive.py:126-144; formula twin
 * extended_mgrd_mixture_model.py:60-108): per-row log p(x), float64 arithmetic.
 * x_dev (B, ld) of x_dtype; logp_dev (B) of out_dtype. */
int mg_gmm_log_prob(mg_primitive *prim, const void *x_dev, int x_dtype, int64_t n_samples, int64_t ld,
                    void *logp_dev, int out_dtype);

/* log_likelihood_jac (reference optimization/objective_functions.py:95-107, inlined again at :190-206):
 * sum_k N_k(x) w_k Sigma_k^-1 (x - mu_k) / p(x) = -grad log p(x), float64; rows where the reference's
 * denominator exp(score(x)) underflows to 0 are all ones, as there.  jac_dev (n_samples, n_components). */
int mg_gmm_log_prob_jac(mg_primitive *prim, const void *x_dev, int x_dtype, int64_t n_samples, int64_t ld,
                        double *jac_dev);

/* GaussianMixture.sample on the device (reference motion_primitive.py:182-189):
 * Philox4x32-10 + Box-Muller + Cholesky; rows grouped by component like sklearn, but NOT
 * bit-compatible with sklearn's Mersenne stream (validated distributionally).  The standard normals
 * carry float32 precision (the uniforms carry 32 bits; Box-Muller on the float32 transcendental units),
 * x = mu + z L^T is float64 arithmetic; same seed, same rows -> same bits on gfx950.
 * counts: host array (K) of rows per component summing to n_samples.
 * x_dev (n, ld) of x_dtype, component_dev (n) int32 or NULL. */
int mg_gmm_sample(mg_primitive *prim, int64_t n_samples, const int64_t *counts, uint64_t seed,
                  void *x_dev, int x_dtype, int64_t ld, int32_t *component_dev);
/* Rows [row_begin, row_begin + row_count) of that same draw of n_samples rows, bit for bit (the generator is counter based: a
 * row's values depend on (seed, row, column group) only): x_dev (row_count, ld), component_dev (row_count).  What a rank of a
 * sharded step draws: the union over the ranks' contiguous blocks IS the single-GPU draw. */
int mg_gmm_sample_rows(mg_primitive *prim, int64_t n_samples, const int64_t *counts, uint64_t seed, int64_t row_begin, int64_t row_count,
                       void *x_dev, int x_dtype, int64_t ld, int32_t *component_dev);

/* ---- fused candidate scoring ------------------------------------------------------------
 * MotionPrimitiveConstraints.evaluate summed over root-joint keyframe constraints
 * (reference motion_primitive_constraints.py:100-122, local-coordinate mode: no
 * alignment), float64 arithmetic, without materialising frames. */
int mg_constraint_set_create(mg_primitive *prim, const mg_keyframe_constraint *cons, int32_t n,
                             mg_constraint_set **out);
/* the same with a skeleton, which MG_CONSTRAINT_JOINT_POSITION constraints need (chains of <= MG_MAX_CHAIN joints) */
int mg_constraint_set_create_fk(mg_primitive *prim, const mg_skeleton_desc *skeleton,
                                const mg_keyframe_constraint *cons, int32_t n, mg_constraint_set **out);
/* global-coordinate mode: every candidate is aligned to the previous motion before its constraints are evaluated
 * (alignment may be NULL = local mode; skeleton may be NULL when no constraint and no alignment needs a chain) */
int mg_constraint_set_create_aligned(mg_primitive *prim, const mg_skeleton_desc *skeleton,
                                     const mg_keyframe_constraint *cons, int32_t n,
                                     const mg_alignment_desc *alignment, mg_constraint_set **out);
/* New targets, weights and reference vectors -- and a new previous frame for the alignment -- for a set whose
 * STRUCTURE stays the same: the same types, joints, keyframes and relative points in the same order, alignment to
 * the same joint (or none).  That is a planner's situation: every step it scores the same kind of constraints with
 * new goals and a new previous motion (graph_walk_planner.py:155-214).  Stream ordered (launches scored before
 * the call see the old values, later ones the new), one small launch, no allocation, no synchronisation;
 * building a new set costs about 200 us, this a few.  MG_ERR_INVALID_ARGUMENT if the structure differs. */
int mg_constraint_set_update(mg_constraint_set *cs, const mg_keyframe_constraint *cons, int32_t n,
                             const mg_alignment_desc *alignment);
/* the general form: MG_CONSTRAINT_POSE entries of `cons` refer to poses[cons[c].joint]; keyframe and weight come
 * from the mg_keyframe_constraint as for every other type.  (mg_constraint_set_update refuses sets with poses.) */
int mg_constraint_set_create_full(mg_primitive *prim, const mg_skeleton_desc *skeleton,
                                  const mg_keyframe_constraint *cons, int32_t n,
                                  const mg_pose_constraint *poses, int32_t n_poses,
                                  const mg_alignment_desc *alignment, mg_constraint_set **out);
void mg_constraint_set_destroy(mg_constraint_set *cs);
int mg_score_constraints(mg_primitive *prim, const mg_constraint_set *cs,
                         const void *latents_dev, int latent_dtype, int64_t n_samples, int64_t ld,
                         void *errors_dev, int out_dtype);

/* The optimiser's objective for a batch in ONE launch: obj[b] = error_scale * err[b] + quality_scale * (-log p(s_b))
 * (reference optimization/objective_functions.py:163-185, obj_spatial_error_sum_and_naturalness; err = what mg_score_constraints
 * returns, log p = what mg_gmm_log_prob returns in float64 -- the same bits: the wave that holds a 16-candidate tile's latents for
 * the mixture scores the constraints on them, and the products and the sum are rounded one by one like NumPy's array arithmetic).
 * logp_out / err_out / obj_out: (n) float64 device pointers, any of them NULL.  Covers sets of root position / 2-D direction
 * constraints (path following), local or aligned through the root joint; MG_ERR_UNSUPPORTED for anything else -- joint
 * constraints, a mixture over time latents, more than 64 dimensions: the kernel runs four waves per SIMD and has no registers for
 * forward-kinematics chains -- then two calls (mg_score_constraints, mg_gmm_log_prob) give the same numbers. */
int mg_objective_error_and_naturalness(mg_primitive *prim, const mg_constraint_set *cs, const void *latents, int dtype, int64_t n, int64_t ld,
                                       double error_scale, double quality_scale, double *logp_out, double *err_out, double *obj_out);

/* MotionPrimitiveConstraints.get_residual_vector (reference motion_primitive_constraints.py:124-144) for the
 * same root-joint keyframe constraints: residuals_dev (n_samples, n_constraints) float64 row-major, entry
 * [b][c] = weight_c * error_c(sample b); feeds the batched forms of obj_spatial_error_residual_vector[_and_
 * naturalness] (reference optimization/objective_functions.py:209-267). */
int mg_score_constraint_residuals(mg_primitive *prim, const mg_constraint_set *cs, const void *latents_dev,
                                  int latent_dtype, int64_t n_samples, int64_t ld, double *residuals_dev);

/* The same matrix with EVERY candidate aligned to ITS OWN previous motion: align_cand_dev (n_samples, 4) float64 = per candidate
 * the previous unit heading (x, z) and the previous root position (x, z), in place of the one record the set was made with (the
 * set must have a previous-frame alignment: its node, chain and reference vector are used).  The steps of a graph walk chained on
 * the device: step i's MG_CONSTRAINT_VALUE_* columns are step i + 1's align_cand (obj_global_error_sum and its residual-vector
 * forms, reference optimization/objective_functions.py:290-380, for a whole batch of concatenated latent vectors). */
int mg_score_constraint_residuals_chained(mg_primitive *prim, const mg_constraint_set *cs, const void *latents_dev, int latent_dtype,
                                          int64_t n_samples, int64_t ld, const double *align_cand_dev, double *residuals_dev);

/* The argmin rule of evaluate_samples_using_constraints
 * (reference motion_primitive_generator.py:251-257): FIRST strict minimum, NaN never
 * wins, (0, +inf) when nothing wins.  values_dev (n) of dtype; result written to host. */
int mg_argmin_first(mg_context *ctx, const void *values_dev, int dtype, int64_t n,
                    int64_t *best_index, double *min_value);
/* device-side result for use inside a stream: out_dev = {int64 index, float64 value} (16 bytes) */
int mg_argmin_first_dev(mg_context *ctx, const void *values_dev, int dtype, int64_t n, void *out_dev);

/* One bench "step": frames (float32, MFMA path) + log p(x) (float32) for the same
 * latent batch, enqueued back to back on the context's stream. */
int mg_step_frames_and_logp(mg_primitive *prim, const void *latents_dev, int latent_dtype,
                            int64_t n_samples, int64_t ld, float *frames_dev, float *logp_dev);

/* What mg_step_frames_and_logp would launch for n_samples candidates, without launching (for profiles and logs):
 * plan[0] = frames kernel (0 = the plain VALU kernel, 1 = tile-major LDS-staged, 2 = chunk-stationary LDS-staged),
 * plan[1] = 1 when log p(x) is scored inside the frames kernel (one launch per step), 0 when it is a second launch,
 * plan[2] = workgroups, plan[3] = LDS bytes per workgroup. */
int mg_step_plan(const mg_primitive *prim, int64_t n_samples, int32_t plan[4]);
/* The same for a given output buffer: where the frames go can change the kernel -- a piece of a placed region whose scan found
 * only slow-class memory (mg_device_malloc) is written by the tile-major kernel, which is the faster one there (frames_dev NULL:
 * mg_step_plan). */
int mg_step_plan_for(const mg_primitive *prim, int64_t n_samples, const void *frames_dev, int32_t plan[4]);

/* evaluate_samples_using_constraints (reference motion_primitive_generator.py:230-261) in one call: score all
 * candidates against the set, first-minimum argmin, result on the host (no allocation, one synchronisation). */
int mg_best_candidate(mg_primitive *prim, const mg_constraint_set *cs, const void *latents_dev, int latent_dtype,
                      int64_t n_samples, int64_t ld, int64_t *best_index, double *min_error);

/* One outgoing option of a planner step (reference graph_walk_planner.py:184-226), enqueued WITHOUT
 * synchronisation: n candidates from the device sampler (counts per mixture component like mg_gmm_sample) into
 * x_dev (n, ld), their errors into errors_dev (n) float64, and into result_dev the first-minimum
 * {int64 index, float64 error} followed by the winning latent as float64[n_components].  Enqueue all options of a
 * step, synchronise once, read the results. */
int mg_option_step(mg_primitive *prim, const mg_constraint_set *cs, int64_t n_samples, const int64_t *counts,
                   uint64_t seed, void *x_dev, int x_dtype, int64_t ld, double *errors_dev, void *result_dev);

/* ALL outgoing options of a planner step (reference graph_walk_planner.py:184-226: the loop over `options`) in one call:
 * mg_option_step for option k with prims[k], csets[k], counts[k], seeds[k], x_dev[k] (n, ld[k]), errors_dev[k]; result
 * record k at results_dev + k * result_stride bytes (result_stride >= 16 + 8 * n_gmm_dims of every option, a multiple
 * of 8).  All primitives must live in one context.  results_host != NULL: the n_options * result_stride bytes are
 * copied back and the stream is synchronised once -- one round trip for the whole step. */
int mg_options_step(int32_t n_options, mg_primitive *const *prims, const mg_constraint_set *const *csets, int64_t n_samples,
                    const int64_t *const *counts, const uint64_t *seeds, void *const *x_dev, int x_dtype, const int64_t *ld,
                    double *const *errors_dev, void *results_dev, int64_t result_stride, void *results_host);

/* mg_options_step with the COMPONENT COUNTS DRAWN ON THE DEVICE: what GaussianMixture.sample draws with
 * numpy.random.multinomial(n, weights) (reference motion_primitive.py:182-189; 4 us of host time per option) becomes the histogram
 * of n categorical draws per option, keyed by the option's seed:
 *     u_i = (Philox4x32-10(counter = (i >> 2, 0, 0, 0x636e7473), key = seed)[i & 3] + 0.5) / 2^32,   i = 0 .. n_samples - 1
 *     counts[c] = #{i : cum[c-1] <= u_i < cum[c]}, cum = cumulative normalised weights (float64), the last component takes the rest.
 * Distributed like NumPy's counts, NOT NumPy's stream (the status of the device sampler itself); given the counts the step is the one
 * mg_options_step makes, bit for bit.  Two launches per step -- ONE when the step before, run with every seed one less, drew this step's
 * counts ahead (its kernel always draws for seed + 1: a planner counts its steps) -- and no host work per option; the kernels leave
 * the result records (and the counts) in pinned host memory themselves, so results_host / counts_host cost one synchronisation and no copy.
 * counts_host (may be NULL): [n_options][16] int64.  At most 24 options; MG_ERR_UNSUPPORTED where an option does not run on the
 * one-launch kernel (more than 16 components, more than 64 mixture dimensions, a VALU kernel forced): draw on the host then. */
int mg_options_step_device_counts(int32_t n_options, mg_primitive *const *prims, const mg_constraint_set *const *csets, int64_t n_samples,
                                  const uint64_t *seeds, void *const *x_dev, int x_dtype, const int64_t *ld,
                                  double *const *errors_dev, void *results_dev, int64_t result_stride, void *results_host, int64_t *counts_host);

/* One rank's share of a step sharded over GPUs (contiguous candidate blocks, constants replicated): the same
 * calls restricted to the global rows [row_begin, row_begin + row_count) of every option's draw of n_samples candidates.
 * x_dev (row_count, ld) and errors_dev (row_count) hold the block; the index in a result record is the GLOBAL row, so the
 * records of all ranks combine by (smaller error, then smaller index) into exactly the single-GPU result. */
int mg_option_step_rows(mg_primitive *prim, const mg_constraint_set *cs, int64_t n_samples, const int64_t *counts, uint64_t seed,
                        int64_t row_begin, int64_t row_count, void *x_dev, int x_dtype, int64_t ld, double *errors_dev, void *result_dev);
int mg_options_step_rows(int32_t n_options, mg_primitive *const *prims, const mg_constraint_set *const *csets, int64_t n_samples,
                         const int64_t *const *counts, const uint64_t *seeds, int64_t row_begin, int64_t row_count,
                         void *const *x_dev, int x_dtype, const int64_t *ld, double *const *errors_dev, void *results_dev,
                         int64_t result_stride, void *results_host);

/* ---- host-pointer convenience variants (H2D, launch, D2H, synchronise) ---------------- */
int mg_back_project_frames_host(mg_primitive *prim, const mg_time_grid *grid, const void *latents,
                                int latent_dtype, int64_t n_samples, int64_t ld, float *frames, int path);
int mg_back_project_frames_f64_host(mg_primitive *prim, const mg_time_grid *grid, const void *latents,
                                    int latent_dtype, int64_t n_samples, int64_t ld, double *frames);
int mg_back_project_coeffs_host(mg_primitive *prim, const void *latents, int latent_dtype,
                                int64_t n_samples, int64_t ld, void *coeffs, int out_dtype);
int mg_spline_evaluate_host(mg_primitive *prim, const mg_time_grid *grid, const double *coeffs,
                            int64_t n_splines, double *frames);
int mg_gmm_log_prob_host(mg_primitive *prim, const void *x, int x_dtype, int64_t n_samples, int64_t ld,
                         void *logp, int out_dtype);
int mg_gmm_sample_host(mg_primitive *prim, int64_t n_samples, const int64_t *counts, uint64_t seed,
                       void *x, int x_dtype, int64_t ld, int32_t *component);
int mg_score_constraints_host(mg_primitive *prim, const mg_constraint_set *cs, const void *latents,
                              int latent_dtype, int64_t n_samples, int64_t ld, void *errors, int out_dtype);
int mg_best_candidate_host(mg_primitive *prim, const mg_constraint_set *cs, const void *latents, int latent_dtype,
                           int64_t n_samples, int64_t ld, int64_t *best_index, double *min_error);
int mg_score_constraint_residuals_host(mg_primitive *prim, const mg_constraint_set *cs, const void *latents,
                                       int latent_dtype, int64_t n_samples, int64_t ld, double *residuals);
int mg_gmm_log_prob_jac_host(mg_primitive *prim, const void *x, int x_dtype, int64_t n_samples, int64_t ld,
                             double *jac);

#ifdef __cplusplus
}
#endif
#endif /* MG_HIP_H */
