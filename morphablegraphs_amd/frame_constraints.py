"""Constraints that walk a joint's position through EVERY frame of a candidate's motion (reference
constraints/spatial_constraints/): trajectory constraints on joints other than the root, trajectory sets, discrete and local
trajectories, collision-avoidance position constraints -- and the local joint-rotation constraint, which reads one frame.

The fused keyframe scorer never materialises frames; these constraints need them, and everything about them happens on the
device, for the whole batch at once:
  mg_back_project_frames_f64   float64 frames of every candidate at the times the constraint reads
  mg_score_constraint_residuals + mg_align_frames
                               the candidate's own aligning transform (rotation about y, xz translation: the closed form the
                               fused scorer applies), derived from its first control point and applied to its frames
  mg_joint_positions           forward kinematics of the wanted joints in every frame -> (n, T, joints, 3) tracks
  mg_score_trajectory_points   TrajectoryConstraint on any joint: the monotone closest-point search over a track
  mg_score_frame_constraint    the other classes: running minimum / arc length / averages, one lane per candidate
Only the (n,) errors and, when asked for, the residual vectors come back.

Device forms (what candidate_scoring.constraints_to_device_form makes of the reference objects):
  {"type": "frame_joint_trajectory", "joint", "control_points", "min_u", "granularity", "weight"}
        TrajectoryConstraint on any joint -- trajectory_constraint.py:79-121
  {"type": "frame_ca_position", "joint", "target": [x | None, ...], "n_frames", "weight"}
        GlobalTransformCAConstraint -- keyframe_constraints/global_transform_ca_constraint.py:33-46
  {"type": "frame_discrete_trajectory", "joint", "points" (m, 3), "unconstrained": [axes], "weight"}
        DiscreteTrajectoryConstraint -- discrete_trajectory_constraint.py:66-90
  {"type": "frame_local_trajectory", "joint", "control_points", "granularity", "start_t", "n_frames", "weight"}
        LocalTrajectoryConstraint -- keyframe_constraints/local_trajectory_constraint.py:45-78
  {"type": "frame_trajectory_set", "joints", "trajectories": [{"control_points", "granularity", "range_start", "range_end"}],
   "arc_lengths", "n_frames", "weight"}
        TrajectorySetConstraint -- trajectory_set_constraint.py:82-104
  {"type": "frame_joint_rotation", "joint_index", "quaternion" (w, x, y, z), "frame_idx", "weight"}
        JointRotationConstraint -- keyframe_constraints/joint_rotation_constraint.py:55-72

PARITY UNPINNED where forward kinematics and alignment are (anim_utils); the target splines and their arc-length look-up are
pinned (tests/golden/trajectory_spline.npz).  There is no CPU fallback: without the library and a GPU every call raises.
"""
import ctypes as C

import numpy as np

from . import _capi

FRAME_TYPES = ("frame_joint_trajectory", "frame_ca_position", "frame_discrete_trajectory", "frame_local_trajectory",
               "frame_trajectory_set", "frame_joint_rotation")


def is_frame_constraint(c):
    return isinstance(c, dict) and c.get("type") in FRAME_TYPES


def split_frame_constraints(clist):
    """(the constraints the fused scorers take, the per-frame ones) of a device-form list"""
    return [c for c in clist if not is_frame_constraint(c)], [c for c in clist if is_frame_constraint(c)]


_ROOT_ONLY = None


def _skeleton_or_root(skeleton):
    global _ROOT_ONLY
    if skeleton is not None:
        return skeleton
    if _ROOT_ONLY is None:
        _ROOT_ONLY = _capi.Skeleton([("root", None, (0.0, 0.0, 0.0))], ["root"])
    return _ROOT_ONLY


class _Batch(object):
    """The candidates of one scoring call on the device, and what has been derived from them so far: aligned float64 frames per
    set of times, joint tracks per (times, joints).  close() frees everything."""

    def __init__(self, prim, S, skeleton, alignment):
        self.prim, self.ctx, self.lib = prim, prim.ctx, prim.lib
        self.S = _capi._latents(S)
        self.n = len(self.S)
        self.skeleton, self.alignment = skeleton, alignment
        self.d_S = self.ctx.upload(self.S)
        self._frames, self._tracks, self._vals, self._buffers = {}, {}, None, [self.d_S]

    def close(self):
        for b in self._buffers:
            b.free()
        self._buffers = []

    def _malloc(self, n_bytes):
        b = self.ctx.malloc(max(int(n_bytes), 8))
        self._buffers.append(b)
        return b

    def _alignment_values(self):
        """(n, 2 | 4) on the device: every candidate's root position (x, z) [and heading (x, z)] in its FIRST control point (a clamped
        spline's value at t = 0) -- what its aligning transform is derived from"""
        if self._vals is None:
            al = self.alignment
            joint = al.get("joint", 0)
            probes = [{"type": "value_position", "t": 0.0, "weight": 1.0, "axis": 0}, {"type": "value_position", "t": 0.0, "weight": 1.0, "axis": 2}]
            if joint != _capi.MG_ALIGN_START_POSE:
                ref_dir = tuple(al.get("ref_dir", (0.0, 0.0, 1.0)))
                probes += [{"type": "value_heading", "t": 0.0, "weight": 1.0, "axis": 0, "joint": joint, "ref_dir": ref_dir},
                           {"type": "value_heading", "t": 0.0, "weight": 1.0, "axis": 2, "joint": joint, "ref_dir": ref_dir}]
            from .candidate_scoring import cached_constraint_set
            cset = cached_constraint_set(self.prim, probes, self.skeleton, None)
            d_v = self._malloc(self.n * len(probes) * 8)
            _capi._check(self.lib.mg_score_constraint_residuals(self.prim.handle, cset.handle, self.d_S.ptr, _capi._dtype_code(self.S), self.n,
                                                                self.S.shape[1], d_v.ptr))
            self._vals = (d_v, len(probes))
        return self._vals

    def frames(self, times):
        """(device buffer (n, T, n_dim) float64, T): the candidates' frames at `times` (canonical times; None = the canonical grid of
        get_motion_vector()), aligned like the candidates"""
        key = None if times is None else tuple(float(t) for t in times)
        if key not in self._frames:
            prim = self.prim
            grid = None if times is None else prim.time_grid(np.asarray(times, dtype=np.float64))
            try:
                T = prim._grid_size(grid)
                d_f = self._malloc(self.n * T * prim.n_dim * 8)
                _capi._check(self.lib.mg_back_project_frames_f64(prim.handle, prim._grid_handle(grid), self.d_S.ptr, _capi._dtype_code(self.S), self.n,
                                                                 self.S.shape[1], d_f.ptr))
            finally:
                if grid is not None:
                    self.ctx.synchronize()      # the launch reads the grid's tables
                    grid.close()
            if self.alignment is not None:
                d_v, nv = self._alignment_values()
                al = _capi.ConstraintSet._marshal_alignment(self.alignment, self.skeleton)
                _capi._check(self.lib.mg_align_frames(prim.handle, d_f.ptr, self.n, T, d_v.ptr, nv, C.byref(al)))
            self._frames[key] = (d_f, T)
        return self._frames[key]

    def tracks(self, joints, times=None):
        """(device buffer (n, T, len(joints), 3) float64, T): the joints' global positions in every frame"""
        sk = _skeleton_or_root(self.skeleton)
        if self.skeleton is None:
            if any(j not in ("root", 0, None) for j in joints):
                raise NotImplementedError("per-frame constraints on joints %r need a skeleton (hip_skeleton); without one only the root's "
                                          "path (joint 0) exists" % (list(joints),))
            joints = [0] * len(joints)
        key = (None if times is None else tuple(float(t) for t in times), tuple(sk.index(j) for j in joints))
        if key not in self._tracks:
            d_f, T = self.frames(times)
            idx = np.ascontiguousarray(key[1], dtype=np.int32)
            d_o = self._malloc(self.n * T * len(idx) * 3 * 8)
            d = sk.desc()
            _capi._check(self.lib.mg_joint_positions(self.ctx.handle, C.byref(d), idx.ctypes.data_as(C.c_void_p), len(idx), d_f.ptr, self.n * T,
                                                     self.prim.n_dim, d_o.ptr))
            self._tracks[key] = (d_o, T)
        return self._tracks[key]


def _score(batch, desc, d_tracks, T, J, d_err, accumulate, want_residuals):
    """one mg_score_frame_constraint launch; the residual vectors (n, m) downloaded when wanted"""
    lib, n = batch.lib, batch.n
    m = lib.mg_frame_constraint_width(C.byref(desc), T)
    d_res = batch._malloc(n * m * 8) if want_residuals else None
    _capi._check(lib.mg_score_frame_constraint(batch.prim.handle, C.byref(desc), d_tracks.ptr, n, T, J, d_err.ptr, 1 if accumulate else 0,
                                               d_res.ptr if d_res is not None else None))
    return batch.ctx.download(d_res, (n, m), np.float64) if want_residuals else None


def _add_frame_constraint(batch, c, d_err, accumulate, want_residuals):
    """Add (or write) constraint c's weighted errors of the batch to d_err (n,) on the device; the residual vectors when wanted."""
    from .candidate_scoring import cached_trajectory
    prim = batch.prim
    kind, w, F = c["type"], float(c.get("weight", 1.0)), prim.n_canonical_frames

    def trajectory_of(t):
        return cached_trajectory(prim, {"type": "trajectory", "control_points": t["control_points"], "granularity": t.get("granularity", 1000)})
    if kind == "frame_joint_trajectory":
        d_tr, T = batch.tracks([c["joint"]])
        d_res = batch._malloc(batch.n * T * 8) if want_residuals else None
        _capi._check(batch.lib.mg_score_trajectory_points(prim.handle, trajectory_of(c).handle, d_tr.ptr, batch.n, T, float(c.get("min_u", 0.0)), w,
                                                          d_err.ptr, 1 if accumulate else 0, d_res.ptr if d_res is not None else None))
        return batch.ctx.download(d_res, (batch.n, T), np.float64) if want_residuals else None
    desc = _capi.FrameConstraintDesc()
    desc.weight, desc.n_joints = w, 1
    keep = []   # what the descriptor points to must outlive the launch
    if kind == "frame_ca_position":
        nf = int(c.get("n_frames", F))
        desc.type, desc.n_frames = _capi.MG_FRAME_CA_POSITION, nf
        for a in range(3):
            t = c["target"][a]
            on = t is not None and not (isinstance(t, float) and np.isnan(t))
            desc.axis_on[a], desc.target[a] = (1, float(t)) if on else (0, 0.0)
        d_tr, T = batch.tracks([c["joint"]], times=np.arange(nf, dtype=np.float64))      # aligned_spline.evaluate(i), i = 0 .. n_canonical_frames - 1
        return _score(batch, desc, d_tr, T, 1, d_err, accumulate, want_residuals)
    if kind == "frame_discrete_trajectory":
        pts = np.ascontiguousarray(np.asarray(c["points"], dtype=np.float64).reshape(-1, 3))
        d_p = batch.ctx.upload(pts) if len(pts) else None
        if d_p is not None:
            batch._buffers.append(d_p)
        desc.type, desc.n_points, desc.points_dev = _capi.MG_FRAME_DISCRETE_TRAJECTORY, len(pts), (d_p.ptr.value if d_p is not None else None)
        free = set(int(a) for a in (c.get("unconstrained") or ()))
        for a in range(3):
            desc.axis_on[a] = 0 if a in free else 1
        d_tr, T = batch.tracks([c["joint"]])
        return _score(batch, desc, d_tr, T, 1, d_err, accumulate, want_residuals)
    if kind == "frame_local_trajectory":
        nf = int(c.get("n_frames", F))
        desc.type, desc.n_frames, desc.start_arc = _capi.MG_FRAME_LOCAL_TRAJECTORY, nf, float(c.get("start_t", 0.0))
        keep.append(trajectory_of(c))
        desc.trajectories[0] = keep[-1].handle.value
        d_tr, T = batch.tracks([c["joint"]], times=np.arange(nf, dtype=np.float64))
        return _score(batch, desc, d_tr, T, 1, d_err, accumulate, want_residuals)
    if kind == "frame_trajectory_set":
        joints = list(c["joints"])
        if not 1 <= len(joints) <= _capi.MG_FRAME_MAX_JOINTS or len(c["trajectories"]) != len(joints):
            raise ValueError("a trajectory set takes 1..%d joints with one trajectory each" % _capi.MG_FRAME_MAX_JOINTS)
        desc.type, desc.n_frames, desc.n_joints = _capi.MG_FRAME_TRAJECTORY_SET, int(c.get("n_frames", F)), len(joints)
        arcs = c.get("arc_lengths", [0.0] * len(joints))
        for j, t in enumerate(c["trajectories"]):
            keep.append(trajectory_of(t))
            desc.trajectories[j] = keep[-1].handle.value
            desc.arc0[j] = float(arcs[j])
            rs, re = t.get("range_start"), t.get("range_end")
            desc.has_range[j] = 0 if rs is None else 1
            desc.range_start[j], desc.range_end[j] = (0.0, 0.0) if rs is None else (float(rs), float(re))
        d_tr, T = batch.tracks(joints)                                                     # the frames of get_motion_vector()
        return _score(batch, desc, d_tr, T, len(joints), d_err, accumulate, want_residuals)
    if kind == "frame_joint_rotation":
        desc.type, desc.quat_channel = _capi.MG_FRAME_JOINT_ROTATION, 3 + 4 * int(c["joint_index"])
        for e in range(4):
            desc.quaternion[e] = float(c["quaternion"][e])
        d_f, T = batch.frames([float(c["frame_idx"])])                                     # aligned_spline.evaluate(frame_idx)
        return _score(batch, desc, d_f, T, prim.n_dim, d_err, accumulate, want_residuals)
    raise ValueError("unknown per-frame constraint %r" % (kind,))


# ---------------------------------------------------------------------------------------------------------------------
# The same constraints WITHOUT frames in memory: mg_joint_tracks (one launch: a workgroup per candidate keeps the control points of
# the channels the joints' chains read in LDS and writes the tracks -- 24 bytes per candidate, time and joint instead of ~98 KB of
# float64 frames per candidate) + mg_score_frame_constraints (the whole list in one launch).  Bit for bit the tracks and errors of
# the chain above, which stays as the route for what the plan does not cover (the local joint-rotation constraint, which reads a
# frame, not a track; more than four distinct (times, joints) requests) and as the oracle-checked reference of the tests.
# ---------------------------------------------------------------------------------------------------------------------
_PLAN_CACHE = {}     # (primitive, skeleton serial, aligning joint, requests) -> _capi.TrackPlan
_GRID_CACHE = {}     # (primitive, n) -> TimeGrid over arange(n)
FUSED = True         # tests switch the fused route off to compare the two


def _request_of(prim, c):
    """(times key, joints) constraint c reads: None = the canonical grid of get_motion_vector(), n = frames 0 .. n-1"""
    kind, F = c["type"], prim.n_canonical_frames
    if kind in ("frame_ca_position", "frame_local_trajectory"):
        return int(c.get("n_frames", F)), (c["joint"],)
    if kind == "frame_trajectory_set":
        return None, tuple(c["joints"])
    return None, (c["joint"],)          # frame_joint_trajectory, frame_discrete_trajectory


def _integer_grid(prim, n):
    key = (prim.serial, prim.handle.value, n)
    g = _GRID_CACHE.get(key)
    if g is None or not g.handle:
        if len(_GRID_CACHE) > 64:
            _GRID_CACHE.clear()
        g = _GRID_CACHE[key] = prim.time_grid(np.arange(n, dtype=np.float64))
    return g


class TrackScorer(object):
    """A list of per-frame constraints (no joint-rotation ones) made ready for batches on the device: the track plan, the grids, the
    constraint records and the trajectories they point to.  score_dev() is the two launches; the track buffers are kept between
    calls of the same batch size.  Raises NotImplementedError when the list is not covered (the caller takes the chain)."""

    def __init__(self, prim, track_list, skeleton, alignment):
        from .candidate_scoring import cached_trajectory as _cached

        def cached_trajectory(p, c):
            # pinned: this scorer's records carry the trajectory's raw handle for as long as the scorer lives (close() releases)
            t = _cached(p, c, pin=True)
            self.keep.append(t)
            return t
        self.keep, self._owned, self._tracks, self._tracks_n = [], [], None, -1
        if not track_list or any(c["type"] == "frame_joint_rotation" for c in track_list):
            raise NotImplementedError("joint-rotation constraints read a frame, not a track")
        sk = _skeleton_or_root(skeleton)
        if skeleton is None and any(j not in ("root", 0, None) for c in track_list for j in _request_of(prim, c)[1]):
            raise NotImplementedError("per-frame constraints on other joints than the root need a skeleton")
        al_joint = 0
        if alignment is not None:
            j = alignment.get("joint", 0)
            al_joint = 0 if j == _capi.MG_ALIGN_START_POSE else sk.index(j)
        reqs, req_of = [], []
        for c in track_list:
            tk, joints = _request_of(prim, c)
            key = (tk, tuple(sk.index(j) if skeleton is not None else 0 for j in joints))
            if key not in reqs:
                reqs.append(key)
            req_of.append(reqs.index(key))
        if len(reqs) > _capi.MG_TRACK_MAX_REQUESTS or any(not 1 <= len(k[1]) <= _capi.MG_FRAME_MAX_JOINTS for k in reqs):
            raise NotImplementedError("more than %d distinct (times, joints) requests" % _capi.MG_TRACK_MAX_REQUESTS)
        pkey = (prim.serial, prim.handle.value, sk.serial, al_joint, tuple(k[1] for k in reqs))
        plan = _PLAN_CACHE.get(pkey)
        if plan is None or not plan.handle:
            if len(_PLAN_CACHE) > 64:
                _PLAN_CACHE.clear()       # (not closed here: a live scorer may hold one; the last reference closes it)
            plan = _PLAN_CACHE[pkey] = _capi.TrackPlan(prim, sk, [list(k[1]) for k in reqs], al_joint)
        self.prim, self.ctx, self.plan, self.alignment = prim, prim.ctx, plan, alignment
        self.reqs, self.req_of = reqs, req_of
        self.grids = [None if k[0] is None else _integer_grid(prim, k[0]) for k in reqs]
        self.Ts = [prim._grid_size(g) for g in self.grids]
        ctx, F = self.ctx, prim.n_canonical_frames
        m = self.m = len(track_list)
        descs, widths = (_capi.FrameConstraintDesc * m)(), []
        try:
            self._fill(prim, track_list, descs, widths, cached_trajectory, ctx, F, req_of)
        except Exception:
            self.close()
            raise
        vp = C.c_void_p
        self.descs, self.widths = descs, widths
        self.dptr = (vp * m)(*[C.addressof(descs[i]) for i in range(m)])
        self.tT = (C.c_int32 * m)(*[self.Ts[req_of[i]] for i in range(m)])
        self.tJ = (C.c_int32 * m)(*[len(reqs[req_of[i]][1]) for i in range(m)])
        self.tptr = None

    def _fill(self, prim, track_list, descs, widths, cached_trajectory, ctx, F, req_of):
        for i, c in enumerate(track_list):
            d, T = descs[i], self.Ts[req_of[i]]
            kind = c["type"]
            d.weight, d.n_joints = float(c.get("weight", 1.0)), 1
            if kind == "frame_joint_trajectory":
                t = cached_trajectory(prim, {"type": "trajectory", "control_points": c["control_points"], "granularity": c.get("granularity", 1000)})
                d.type, d.start_arc = _capi.MG_FRAME_JOINT_TRAJECTORY, float(c.get("min_u", 0.0))
                d.trajectories[0] = t.handle.value
            elif kind == "frame_ca_position":
                d.type, d.n_frames = _capi.MG_FRAME_CA_POSITION, int(c.get("n_frames", F))
                for a in range(3):
                    t = c["target"][a]
                    on = t is not None and not (isinstance(t, float) and np.isnan(t))
                    d.axis_on[a], d.target[a] = (1, float(t)) if on else (0, 0.0)
            elif kind == "frame_discrete_trajectory":
                pts = np.ascontiguousarray(np.asarray(c["points"], dtype=np.float64).reshape(-1, 3))
                d_p = ctx.upload(pts) if len(pts) else None
                if d_p is not None:
                    self._owned.append(d_p)
                d.type, d.n_points, d.points_dev = _capi.MG_FRAME_DISCRETE_TRAJECTORY, len(pts), (d_p.ptr.value if d_p is not None else None)
                free = set(int(a) for a in (c.get("unconstrained") or ()))
                for a in range(3):
                    d.axis_on[a] = 0 if a in free else 1
            elif kind == "frame_local_trajectory":
                t = cached_trajectory(prim, {"type": "trajectory", "control_points": c["control_points"], "granularity": c.get("granularity", 1000)})
                d.type, d.n_frames, d.start_arc = _capi.MG_FRAME_LOCAL_TRAJECTORY, int(c.get("n_frames", F)), float(c.get("start_t", 0.0))
                d.trajectories[0] = t.handle.value
            elif kind == "frame_trajectory_set":
                joints = list(c["joints"])
                if len(c["trajectories"]) != len(joints):
                    raise ValueError("a trajectory set takes 1..%d joints with one trajectory each" % _capi.MG_FRAME_MAX_JOINTS)
                d.type, d.n_frames, d.n_joints = _capi.MG_FRAME_TRAJECTORY_SET, int(c.get("n_frames", F)), len(joints)
                arcs = c.get("arc_lengths", [0.0] * len(joints))
                for j, t in enumerate(c["trajectories"]):
                    tr = cached_trajectory(prim, {"type": "trajectory", "control_points": t["control_points"], "granularity": t.get("granularity", 1000)})
                    d.trajectories[j] = tr.handle.value
                    d.arc0[j] = float(arcs[j])
                    rs, re = t.get("range_start"), t.get("range_end")
                    d.has_range[j] = 0 if rs is None else 1
                    d.range_start[j], d.range_end[j] = (0.0, 0.0) if rs is None else (float(rs), float(re))
            else:
                raise ValueError("unknown per-frame constraint %r" % (kind,))
            widths.append(prim.lib.mg_frame_constraint_width(C.byref(d), T))

    def _track_buffers(self, n):
        if self._tracks_n != n:
            self._free_tracks()
            self._tracks = [self.ctx.malloc(max(n, 1) * T * len(k[1]) * 3 * 8) for T, k in zip(self.Ts, self.reqs)]
            self._tracks_n = n
            self.tptr = (C.c_void_p * self.m)(*[self._tracks[self.req_of[i]].ptr.value for i in range(self.m)])
        return self._tracks

    def _free_tracks(self):
        for b in self._tracks or ():
            b.free()
        self._tracks, self._tracks_n = None, -1

    def valid(self):
        """everything the records point to is still alive (a cleared cache or a closed primitive takes it away)"""
        return bool(self.plan.handle) and all(t.handle for t in self.keep) and all(g is None or g.handle for g in self.grids)

    def score_dev(self, lat_dev, lat_dtype, n, ld, d_err, accumulate=True, residual_devs=None):
        """mg_joint_tracks + mg_score_frame_constraints on resident latents: the list's errors added to (or written over) d_err (n,)."""
        if not self.valid():
            raise _capi.MGError("TrackScorer: a trajectory or plan it points to has been closed (its primitive or the caches were cleared)")
        tracks = self._track_buffers(n)
        self.plan.tracks_dev(lat_dev, lat_dtype, n, ld, self.grids, tracks, self.alignment)
        rptr = (C.c_void_p * self.m)(*[_capi._dev_ptr(r).value for r in residual_devs]) if residual_devs is not None else None
        _capi._check(self.prim.lib.mg_score_frame_constraints(self.prim.handle, self.m, self.dptr, self.tptr, self.tT, self.tJ, n, _capi._dev_ptr(d_err),
                                                              1 if accumulate else 0, rptr))

    def close(self):
        if self.ctx.handle:
            self.ctx.synchronize()
            self._free_tracks()
            for b in self._owned:
                b.free()
        self._owned = []
        from .candidate_scoring import release_trajectory
        for t in self.keep:
            release_trajectory(t)
        self.keep = []


def _fused(prim, S, frame_list, skeleton, alignment, d_err, accumulate, residuals):
    """The fused route; None when the list is not covered (the caller takes the chain)."""
    if not FUSED:
        return None
    try:
        scorer = TrackScorer(prim, frame_list, skeleton, alignment)
    except NotImplementedError:
        return None
    S = _capi._latents(S)
    n, ctx = len(S), prim.ctx
    bufs = [ctx.upload(S)]
    try:
        res = [ctx.malloc(max(n, 1) * w * 8) for w in scorer.widths] if residuals else None
        bufs += res or []
        scorer.score_dev(bufs[0], S.dtype, n, S.shape[1], d_err, accumulate, res)
        blocks = [ctx.download(r, (n, w), np.float64) for r, w in zip(res, scorer.widths)] if residuals else []
        return blocks
    finally:
        scorer.close()
        for b in bufs:
            b.free()


def add_frame_constraints_dev(prim, S, frame_list, skeleton, alignment, d_err, accumulate=True, residuals=False):
    """Add the per-frame constraints' weighted errors of candidates S (host latents) to d_err (n,) float64 on the device (accumulate
    False: the first one overwrites).  Returns the list of residual blocks (n, m_c) when residuals is set.
    The additions happen in the LIST's order, as the reference's loop makes them (motion_primitive_constraints.py:117-121): runs
    of consecutive track constraints go through the fused route (two launches per run), a joint-rotation constraint -- which reads
    a frame, not a track -- through the chain at its place in the list (tracks first, rotations last changed the
    float64 summation order of mixed lists)."""
    runs, i = [], 0
    while i < len(frame_list):
        j = i
        if frame_list[i]["type"] != "frame_joint_rotation":
            while j < len(frame_list) and frame_list[j]["type"] != "frame_joint_rotation":
                j += 1
            runs.append((True, list(range(i, j))))
        else:
            j = i + 1
            runs.append((False, [i]))
        i = j
    blocks = [None] * len(frame_list)
    batch, first = None, not accumulate
    try:
        for on_tracks, idx in runs:
            fused = _fused(prim, S, [frame_list[i] for i in idx], skeleton, alignment, d_err, not first, residuals) if on_tracks else None
            if fused is not None:
                if residuals:
                    for i, b in zip(idx, fused):
                        blocks[i] = b
                first = False
                continue
            if batch is None:
                batch = _Batch(prim, S, skeleton, alignment)
            for i in idx:
                blocks[i] = _add_frame_constraint(batch, frame_list[i], d_err, not first, residuals)
                first = False
        if batch is not None:
            batch.ctx.synchronize()
    finally:
        if batch is not None:
            batch.close()
    return blocks if residuals else None


def frame_constraint_residuals(prim, S, c, skeleton=None, alignment=None):
    """The constraint's residual vector for every candidate, times its weight, and what MotionPrimitiveConstraints.evaluate adds
    for it: ((n, m) residuals, (n,) errors)."""
    err, blocks = frame_constraints_errors(prim, S, [c], skeleton, alignment)
    return blocks[0], err


def frame_constraints_errors(prim, S, frame_list, skeleton=None, alignment=None):
    """(n,) the sum of the per-frame constraints' errors, and their residual columns side by side"""
    S = _capi._latents(S)
    n = len(S)
    if n == 0 or not frame_list:
        return np.zeros(n), [np.zeros((n, 0)) for _ in frame_list]
    d_err = prim.ctx.malloc(n * 8)
    try:
        blocks = add_frame_constraints_dev(prim, S, frame_list, skeleton, alignment, d_err, accumulate=False, residuals=True)
        return prim.ctx.download(d_err, (n,), np.float64), blocks
    finally:
        d_err.free()
