"""A minimal container for BASELINE.json config 3: the ~16 primitives of a MotionStateGraph resident on
one GPU, and the batched form of GraphWalkPlanner's option evaluation
(reference morphablegraphs/motion_generator/graph_walk_planner.py:184-226): for every outgoing option draw
n candidates, score them against the path-following constraints, keep the first minimum per option, pick
the option with the smallest error (np.argmin over options, graph_walk_planner.py:191-192).
Graph loading / transitions / control flow stay in the reference; only the scoring is replaced."""
import numpy as np

from . import candidate_scoring as _cs

from . import _capi
from .candidate_scoring import constraints_to_device_form, evaluate_samples_using_constraints
from .frame_constraints import is_frame_constraint
from .motion_primitive import HipMotionPrimitive, get_context
from .motion_primitive_wrapper import HipMotionPrimitiveModelWrapper

NODE_TYPE_START = "start"
NODE_TYPE_STANDARD = "standard"
NODE_TYPE_END = "end"


def arc_length_xz(root_positions):
    """Length of the root path on the ground plane (x, z) -- what the reference gets from anim_utils'
    extract_root_positions_from_frames + get_arc_length_from_points (motion_state_graph_node.py:222-224;
    PARITY UNPINNED: anim_utils is not under /root/reference)."""
    p = np.asarray(root_positions, dtype=np.float64)[..., [0, 2]]
    return np.sqrt(((p[..., 1:, :] - p[..., :-1, :]) ** 2).sum(axis=-1)).sum(axis=-1)


class HipMotionStateGraphNode(HipMotionPrimitiveModelWrapper):
    """The hot-path call sites of MotionStateGraphNode (reference motion_model/motion_state_graph_node.py:45-275)
    on top of the HIP-backed wrapper: step-length statistics, transition-model dispatch, best-sample search.
    Edges / groups / cluster trees are plain host data exactly as in the reference."""

    def __init__(self, motion_state_group=None, context=None, device=0):
        super(HipMotionStateGraphNode, self).__init__(context=context, device=device)
        self.motion_state_group = motion_state_group
        self.outgoing_edges = dict()
        self.node_type = NODE_TYPE_STANDARD
        self.n_standard_transitions = 0
        self.parameter_bb = None
        self.cartesian_bb = None
        self.velocity_data = None
        self.average_step_length = 0
        self.action_name = None
        self.name = None
        self.node_key = None
        self.cluster_tree = None

    def init_from_dict(self, action_name, desc):
        self.name = desc["name"]
        self.action_name = action_name
        self.node_key = (action_name, self.name)       # motion_state_graph_node.py: the key the graph holds the node under
        self._initialize_from_json(None, desc["mm"])
        if "stats" in desc:
            self.parameter_bb = desc["stats"]["pose_bb"]
            self.cartesian_bb = desc["stats"]["cartesian_bb"]
            self.velocity_data = desc["stats"]["pose_velocity"]

    # ---- step length (motion_state_graph_node.py:183-230) ------------------------------------------
    def update_motion_stats(self, n_samples=5, method="median"):
        self.n_standard_transitions = len([e for e in self.outgoing_edges
                                           if getattr(self.outgoing_edges[e], "transition_type", None) == NODE_TYPE_STANDARD])
        sample_lengths = [self._get_random_sample_step_length() for _ in range(n_samples)]
        if method == "average":
            self.average_step_length = sum(sample_lengths) / n_samples
        else:
            self.average_step_length = np.median(sample_lengths)

    def _get_random_sample_step_length(self, method="arc_length"):
        current_parameters = np.ravel(self.sample_low_dimensional_vector())
        return self.get_step_length_for_sample(current_parameters, method)

    def get_step_length_for_sample(self, parameters, method="arc_length"):
        quat_frames = self.back_project(parameters, use_time_parameters=False).get_motion_vector()
        if method == "arc_length":
            return float(arc_length_xz(quat_frames[:, :3]))
        elif method == "distance":
            return float(np.linalg.norm(quat_frames[-1][:3] - quat_frames[0][:3]))
        raise NotImplementedError

    def get_step_lengths_for_samples(self, samples, method="arc_length"):
        """Batched form: one frames launch for all rows."""
        frames = self.motion_primitive.back_project_frames_batch(np.asarray(samples))
        if method == "arc_length":
            return arc_length_xz(frames[:, :, :3])
        elif method == "distance":
            return np.linalg.norm(frames[:, -1, :3].astype(np.float64) - frames[:, 0, :3], axis=1)
        raise NotImplementedError

    # ---- transitions (motion_state_graph_node.py:232-272) -----------------------------------------
    def has_transition_model(self, to_node_key):
        return to_node_key in self.outgoing_edges and getattr(self.outgoing_edges[to_node_key], "transition_model", None) is not None

    def predict_parameters(self, to_node_key, current_parameters):
        gmm = self.outgoing_edges[to_node_key].transition_model.predict(current_parameters)
        s = gmm.sample()           # ONE draw, as motion_state_graph_node.py:252-253: every call consumes NumPy's global stream
        return np.ravel(s[0] if isinstance(s, tuple) else s)

    def predict_gmm(self, to_node_key, current_parameters):
        edge = self.outgoing_edges.get(to_node_key)
        if edge is not None and getattr(edge, "transition_model", None) is not None:
            return edge.transition_model.predict(current_parameters)
        return self.get_gaussian_mixture_model()

    # ---- best-sample search (motion_state_graph_node.py:119-142) -------------------------------------
    def search_best_sample(self, obj, data, n_candidates=2):
        if self.cluster_tree is not None:
            return self.cluster_tree.find_best_example_excluding_search_candidates(obj, data, n_candidates)
        return np.inf, None

    def search_best_sample_gpu(self, constraints, n_samples):
        """Brute-force replacement of the cluster-tree search at GPU batch sizes: draw n_samples latents, score
        them in one launch, return (error, parameters) like search_best_sample."""
        samples = self.sample_low_dimensional_vectors(n_samples)
        best, err = evaluate_samples_using_constraints(samples, self, constraints, None)
        return err, best


class HipMotionStateTransition(object):
    """MotionStateTransition (reference motion_model/motion_state_transition.py): plain host data."""

    def __init__(self, from_node_key, to_node_key, transition_type, transition_model=None):
        self.from_node_key, self.to_node_key = from_node_key, to_node_key
        self.transition_type, self.transition_model = transition_type, transition_model


class _StoredSamples(object):
    """What the brute-force search needs of a FeatureClusterTree: its stored samples."""

    def __init__(self, data):
        self.data = np.asarray(data, dtype=np.float64)


NODE_TYPE_SINGLE = "single_primitive"
NODE_TYPE_CYCLE_END = "cycle_end"
NODE_TYPE_IDLE = "idle"


class HipMotionStateGraph(object):
    """All primitives of a motion state graph resident on one GPU (BASELINE.json config 3), built the way
    MotionStateGraphLoader._build_from_zip_file does (reference motion_model/motion_state_graph_loader.py:184-242):
    node groups from the "subgraphs" of the zip, node types from each action's meta information
    (motion_state_group.py:46-62), transitions from the "transitions" table (loader :244-263, types :265-289),
    the start node, cached or recomputed step statistics.  Skeleton, hand poses and PFNN data are not read."""

    def __init__(self, context=None, device=0):
        self.ctx = context or get_context(device)
        self.nodes = {}
        self.node_groups = {}
        self.start_node = None
        self.action_definitions = {}

    def load_from_zip(self, path, recalculate_stats=False):
        from .model_io import read_graph_zip
        return self.build_from_graph_data(read_graph_zip(path), recalculate_stats)

    def build_from_graph_data(self, graph_data, recalculate_stats=False):
        # every primitive of the graph lives in one device arena (a few 64 MiB blocks instead of ~20 allocations each)
        self.ctx.arena_begin()
        try:
            self._build_nodes(graph_data)
        finally:
            self.ctx.arena_end()
        self._set_transitions_from_dict(graph_data.get("transitions", {}))
        for group in self.node_groups.values():
            stats = group["info"].get("stats", {})
            for mp_name in group["nodes"]:
                node = self.nodes[(group["name"], mp_name)]
                if recalculate_stats or mp_name not in stats:
                    node.update_motion_stats()
                else:   # motion_state_group.py:88-99: cached values from meta_information.json
                    node.average_step_length = stats[mp_name]["average_step_length"]
                    node.n_standard_transitions = stats[mp_name]["n_standard_transitions"]
        if "actionDefinitions" in graph_data:
            self.action_definitions = graph_data["actionDefinitions"]
        if "startNode" in graph_data:
            start_node = list(graph_data["startNode"])
            if start_node[1].startswith("walk"):
                start_node[1] = start_node[1][5:]
            self.start_node = tuple(start_node)
        return self

    def _build_nodes(self, graph_data):
        for action_name, action_data in graph_data["subgraphs"].items():
            group = {"name": action_data["name"], "info": action_data.get("info", {}), "nodes": []}
            for mp_name, desc in action_data["nodes"].items():
                if "spatial_coeffs" in desc["mm"]:
                    continue   # static primitives carry no statistical model (motion_primitive_wrapper.py:61-66)
                node = HipMotionStateGraphNode(group, context=self.ctx)
                node.init_from_dict(action_data["name"], desc)
                if "space_partition_json" in desc:
                    node.cluster_tree = _StoredSamples(desc["space_partition_json"]["data"])
                self.nodes[(action_data["name"], mp_name)] = node
                group["nodes"].append(mp_name)
            self._set_node_types(group)
            self.node_groups[action_data["name"]] = group
            idle = group["info"].get("idle_states", [])
            if action_name == "walk" and len(idle) > 0:
                self.start_node = (action_name, idle[0])

    def _set_node_types(self, group):
        keys = [(group["name"], n) for n in group["nodes"]]
        if len(keys) == 1:
            self.nodes[keys[0]].node_type = NODE_TYPE_SINGLE
            return
        info = group["info"]
        for field, node_type in (("start_states", NODE_TYPE_START), ("end_states", NODE_TYPE_END),
                                 ("cycle_states", NODE_TYPE_CYCLE_END), ("idle_states", NODE_TYPE_IDLE)):
            for k in info.get(field, []):
                if (group["name"], k) in self.nodes:
                    self.nodes[(group["name"], k)].node_type = node_type

    def _get_transition_type(self, from_node_key, to_node_key):
        t_type = "action_transition"
        if to_node_key[0] == from_node_key[0]:
            to_type = self.nodes[to_node_key].node_type
            if self.nodes[from_node_key].node_type == NODE_TYPE_IDLE:
                if to_type in (NODE_TYPE_START, NODE_TYPE_IDLE, NODE_TYPE_END):
                    t_type = to_type
            else:
                t_type = to_type if to_type in (NODE_TYPE_STANDARD, NODE_TYPE_START, NODE_TYPE_CYCLE_END, NODE_TYPE_IDLE) else NODE_TYPE_END
        return t_type

    def _set_transitions_from_dict(self, transition_dict):
        if len(transition_dict) == 0:
            return
        split_key = ":" if ":" in list(transition_dict.keys())[0] else "_"
        for node_key in transition_dict:
            from_node_key = tuple(node_key.split(split_key)[:2])
            if from_node_key not in self.nodes:
                continue
            for to_key in transition_dict[node_key]:
                to_node_key = tuple(to_key.split(split_key)[:2])
                if to_node_key in self.nodes:
                    self.nodes[from_node_key].outgoing_edges[to_node_key] = HipMotionStateTransition(
                        from_node_key, to_node_key, self._get_transition_type(from_node_key, to_node_key), None)

    def close(self):
        for node in self.nodes.values():
            prim = getattr(node.motion_primitive, "_prim", None)
            if prim is not None:
                prim.close()
        self.nodes = {}


def _fp_value(v):
    t = type(v)
    if v is None or t is float or t is int or t is str or t is bool:
        return v
    if t is list or t is tuple:
        return tuple([x if type(x) is float else _fp_value(x) for x in v])
    if t is np.ndarray:
        return (v.dtype.str, v.shape, v.tobytes())
    if isinstance(v, np.generic):
        return v.item()
    if t is dict:
        return tuple([(k, _fp_value(x)) for k, x in v.items()])
    raise TypeError(t)


_FLAT = frozenset((float, int, str, bool, type(None)))
_DTYPES = {np.float32: np.dtype(np.float32), np.float64: np.dtype(np.float64)}


def flat_constraint_copy(clist):
    """A value copy of a list of plain device-form constraint dicts whose every value is a scalar or a flat list of scalars (the
    usual form: targets as lists of floats / None), for the planner step's "same constraints as last step?" test; None for
    anything else -- arrays, nested lists, reference objects -- which takes the general route every step.  The copy shares
    nothing mutable with the caller's dicts: a target rewritten in place changes the comparison's outcome."""
    if type(clist) is not list:
        return None
    out = []
    for c in clist:
        if type(c) is not dict:
            return None
        d = {}
        for k, v in c.items():
            t = type(v)
            if t in _FLAT:
                d[k] = v
            elif t is list:
                for x in v:
                    if type(x) not in _FLAT:
                        return None
                d[k] = v[:]
            else:
                return None
        out.append(d)
    return out


def _same_mapping(cons, remembered):
    """{option: constraint list} == the remembered flat copy, by value, in one comparison."""
    try:
        return cons == remembered
    except (ValueError, TypeError):
        return False


def _same_constraints(clist, remembered):
    """clist == remembered by value (the interpreter's own recursive comparison of lists, dicts and scalars: key names, lengths
    and values; NaN compares unequal, so a NaN target is "changed" every step -- safe).  Values that cannot be compared this way
    (arrays) are "not the same"."""
    try:
        return type(clist) is list and clist == remembered
    except (ValueError, TypeError):
        return False


def constraint_fingerprint(clist):
    """A value copy of a list of plain device-form constraint dicts, for "same as last step?" comparisons: key names and
    values, nested lists and arrays copied element by element (a caller that rewrites a target list or array IN PLACE changes
    the next fingerprint, not the remembered one).  None when the list holds anything else (reference constraint objects,
    values of unknown types): such lists take the general route."""
    if type(clist) is not list:
        return None
    try:
        return [tuple([(k, _fp_value(v)) for k, v in c.items()]) for c in clist]
    except (AttributeError, TypeError):
        return None


def _options_frame_lists(plan, steps, fast, extras, n, dtype):
    """mg_options_frame_lists for the options in `fast` [(k, TrackScorer | None)]: [(index, error, winning latent as float64)]."""
    import ctypes as C
    vp, m = C.c_void_p, len(fast)
    R, Q = _capi.MG_TRACK_MAX_REQUESTS, _capi.MG_FRAME_LIST_MAX
    prims, plans, lats, errs = (vp * m)(), (vp * m)(), (vp * m)(), (vp * m)()
    lds, ncons = (C.c_int64 * m)(), (C.c_int32 * m)()
    grids, tracks, cons = (vp * (m * R))(), (vp * (m * R))(), (vp * (m * Q))()
    req_of = (C.c_int32 * (m * Q))()
    als, al_ptrs = [], (vp * m)()
    ctx = steps[fast[0][0]][3]
    for j, (k, scorer) in enumerate(fast):
        name, node, prim, _, d_x, d_e, d_r, L, pvals = steps[k]
        prims[j], lats[j], errs[j], lds[j] = prim.handle, _capi._dev_ptr(d_x), _capi._dev_ptr(d_e), L
        alignment, sk = extras[k][2], extras[k][3]
        if scorer is not None:
            if not scorer.valid():
                raise _capi.MGError("a track scorer's plan or trajectories were closed under it")
            plans[j], ncons[j] = scorer.plan.handle, scorer.m
            bufs = scorer._track_buffers(n)
            for q, (g, b) in enumerate(zip(scorer.grids, bufs)):
                grids[j * R + q] = g.handle if g is not None else None
                tracks[j * R + q] = _capi._dev_ptr(b)
            for i in range(scorer.m):
                cons[j * Q + i] = C.addressof(scorer.descs[i])
                req_of[j * Q + i] = scorer.req_of[i]
            if alignment is not None:
                al = _capi.ConstraintSet._marshal_alignment(alignment, scorer.plan.skeleton)
                als.append(al)
                al_ptrs[j] = C.addressof(al)
    L_max = max(steps[k][7] for k, _ in fast)
    stride = 16 + 8 * L_max
    key = ("frame_lists_results", m, stride)
    bufs = plan.get(key)
    if bufs is None:
        bufs = plan[key] = (ctx.malloc(m * stride), np.empty(m * stride, dtype=np.uint8))
    d_res, host = bufs
    code = _capi.MG_F64 if np.dtype(dtype) == np.float64 else _capi.MG_F32
    _capi._check(ctx.lib.mg_options_frame_lists(m, prims, plans, lats, code, n, lds, al_ptrs, grids, tracks, ncons, cons, req_of, errs, d_res.ptr, stride,
                                                 host.ctypes.data_as(vp)))
    rec = host.reshape(m, stride)
    out = []
    for j in range(m):
        idx = int(rec[j, 0:8].view(np.int64)[0])
        err = float(rec[j, 8:16].view(np.float64)[0])
        out.append((idx, err, rec[j, 16:].view(np.float64).copy()))
    return out


class HipPrimitiveSet(object):
    """separate_streams: every primitive gets its own libmg_hip context, i.e. its own HIP stream, so that the small,
    latency-bound launches of different options overlap on the GPU (evaluate_options_on_device)."""

    def __init__(self, primitives_json, context=None, device=0, separate_streams=False):
        self.ctx = context or get_context(device)
        self.nodes = {}
        self._buffers = {}
        for data in primitives_json:
            ctx = _capi.Context(device) if separate_streams else self.ctx
            p = HipMotionPrimitive(None, context=ctx)
            p._initialize_from_json(data)
            self.nodes[p.name] = p

    @property
    def last_counts(self):
        """{option: component counts} the device drew in the last step with device_counts=True."""
        raw = getattr(self, "_last_counts", None)
        if raw is None:
            return None
        cnt, steps = raw
        return {st[0]: cnt[k, :len(st[8])].copy() for k, st in enumerate(steps)}

    def _repeat_step(self, plan, options, n, seed, dt, device_counts):
        """evaluate_options_on_device for a step whose constraint sets are the last step's (the caller has checked): the C call and
        the unpacking of the result records, nothing else.  None: the device cannot draw this step's counts (host route)."""
        steps = plan["steps"]
        m, stride, host = len(steps), plan["stride"], plan["host"]
        code = _capi.MG_F64 if dt == np.float64 else _capi.MG_F32
        np.add(plan["karange"], np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), out=plan["seeds_np"])
        if device_counts and m <= 24:
            rc = plan["lib"].mg_options_step_device_counts(m, plan["prims"], plan["csets"], n, plan["seeds"], plan["xs"], code, plan["lds"],
                                                           plan["errs"], plan["shared_ptr"], stride, plan["host_ptr"], plan["device_counts_ptr"])
            if rc == -4:
                return None
            self._last_counts = (plan["device_counts"], steps)
        else:
            counts, multinomial = plan["counts"], np.random.multinomial
            for k, st in enumerate(steps):   # the component counts, in option order, from NumPy's global stream
                counts[k, :len(st[8])] = multinomial(n, st[8])
            rc = plan["lib"].mg_options_step(m, plan["prims"], plan["csets"], n, plan["cnts"], plan["seeds"], plan["xs"], code, plan["lds"],
                                             plan["errs"], plan["shared_ptr"], stride, plan["host_ptr"])
        if rc != 0:
            _capi._check(rc)
        rec = host[:m * stride].copy().reshape(m, stride)          # one copy: the host block is reused by the next step
        errs = rec[:, 8:16].copy().view(np.float64)[:, 0].tolist()
        lat = rec[:, 16:].copy().view(np.float64)                    # (m, widest L): the winners, already rounded to the caller's type
        results = {st[0]: (lat[k, :st[7]], errs[k]) for k, st in enumerate(steps)}
        return options[min(range(m), key=errs.__getitem__)], results      # first minimum (the kernel never reports NaN)

    def evaluate_options_on_device(self, options, constraints_per_option, n_samples, seed=0, dtype=np.float32,
                                   prev_frames=None, skeleton=None, communicator=None, device_counts=False):
        """GraphWalkPlanner's option evaluation (reference graph_walk_planner.py:184-226) without host round trips:
        for every option the component counts come from NumPy's stream, the candidates from the device sampler,
        scoring, first-minimum argmin and the copy of the winner stay on the device (mg_options_step: one C call enqueues
        every option without synchronisation and reads all (16 + 8 L)-byte result records back in one copy; with a
        context per primitive: mg_option_step per option, then one read-back each).
        With `prev_frames` every candidate is aligned to the last previous frame before scoring, which is how the
        planner scores (its constraints stay global, graph_walk_planner.py:179; `skeleton`: a _capi.Skeleton when
        the aligning node is not the root or joints other than the root are constrained).
        communicator (distributed.MgCommunicator / FileCommunicator; rank 0 calls, the other ranks sit in
        distributed.worker_loop with their own HipPrimitiveSet under "__primitive_set__"): rank 0 draws the counts and broadcasts
        them with the constraint values and the seed, every rank runs the step on its block of the rows of every option's draw
        (mg_options_step_rows), one all-gather of the result records, per option the first minimum over the ranks: the
        single-GPU result.
        device_counts: the component counts of every option's draw come from the device as well (mg_options_step_device_counts:
        a Philox-keyed multinomial per option, distributed like NumPy's, not NumPy's stream -- the status the device sampler has
        anyway) instead of 16 x np.random.multinomial on the host, which is most of a step's host time; steps the one-launch
        kernel does not cover fall back to the host draw.  self.last_counts holds the counts of the last such step.
        Returns (best_option, {name: (best_sample, min_error)})."""
        n = int(n_samples)
        # The step a planner repeats: same options, same batch, same plain constraint values as last time, nothing in between
        # (see "The whole step's shortcut" below) -- everything the C call needs is in the plan, made once.
        dt = _DTYPES.get(dtype) or np.dtype(dtype)
        if communicator is None and prev_frames is None and skeleton is None and type(constraints_per_option) is dict:
            plan = self._buffers.get(("plan", options if type(options) is tuple else tuple(options), n, dt.str))
            whole = plan.get("whole") if plan is not None else None
            if whole is not None and whole[1] == _cs.CSET_GENERATION[0] and len(constraints_per_option) == len(whole[0]) and \
                    plan["one_context"] and _same_mapping(constraints_per_option, whole[0]):
                out = self._repeat_step(plan, options, n, seed, dt, device_counts)
                if out is not None:
                    return out
        cached_constraint_set, alignment_from_prev_frames, CSET_GENERATION = _cs.cached_constraint_set, _cs.alignment_from_prev_frames, _cs.CSET_GENERATION
        import ctypes as C
        if communicator is not None and communicator.world > 1:
            from . import distributed
            cmd = {"op": "options_step", "options": list(options), "n_samples": n, "seed": int(seed), "dtype": np.dtype(dtype).name,
                   "skeleton": skeleton is not None, "counts": {}, "constraints": {}, "alignments": {}, "widths": {}}
            for name in options:
                node = self.nodes[name]
                cons = constraints_per_option[name]
                clist = cons.constraints if hasattr(cons, "constraints") else cons
                sk = skeleton if skeleton is not None else getattr(cons, "hip_skeleton", None)
                w = np.asarray(node.gaussian_mixture_model.weights_, dtype=np.float64)
                cmd["counts"][name] = np.random.multinomial(n, w / w.sum()).astype(np.int64)
                cmd["constraints"][name] = constraints_to_device_form(clist)
                cmd["alignments"][name] = alignment_from_prev_frames(prev_frames, cons, sk)
                cmd["widths"][name] = node._prim.n_gmm_dims
            local = {"__primitive_set__": self, "__skeleton__": skeleton}
            if any(is_frame_constraint(c) or c.get("type") == "trajectory" for name in options for c in cmd["constraints"][name]):
                # an option with a trajectory or per-frame constraint: the step as one sample-and-evaluate command per option (every
                # rank holds the primitives under their names), the same draws
                local.update(self.nodes)
                out = {}
                for k, name in enumerate(options):
                    _, err, lat = distributed.run_command(communicator, local, {
                        "op": "sample_and_evaluate", "node": name, "constraints": cmd["constraints"][name], "alignment": cmd["alignments"][name],
                        "skeleton": cmd["skeleton"], "counts": cmd["counts"][name], "seed": int(seed) + k, "dtype": cmd["dtype"],
                        "width": cmd["widths"][name]})
                    out[name] = (np.asarray(lat, dtype=np.float64), err)
            else:
                out = distributed.run_command(communicator, local, cmd)
            results = {name: (out[name][0].astype(dtype).astype(np.float64), out[name][1]) for name in options}
            errors = [results[nm][1] for nm in options]
            return options[int(np.argmin(errors))], results
        code = _capi.MG_F64 if np.dtype(dtype) == np.float64 else _capi.MG_F32
        plan = self._step_plan(tuple(options), n, np.dtype(dtype))
        steps = plan["steps"]
        csets, general = [], []
        memo = plan.setdefault("memo", [None] * len(steps))
        multinomial = np.random.multinomial
        # The whole step's shortcut.  A planner that asks the same questions as at the last step -- every option's constraints plain
        # device-form dicts with the values they had (ONE comparison by value of the whole mapping against a copy that shares nothing
        # mutable with the caller's: a target rewritten in place is seen), no shared set created, rewritten or closed since (one
        # integer) -- scores against the sets the last step used: no per-option host work at all.
        whole = plan.get("whole")
        if whole is not None and prev_frames is None and skeleton is None and whole[1] == CSET_GENERATION[0] and \
                type(constraints_per_option) is dict and len(constraints_per_option) == len(whole[0]) and _same_mapping(constraints_per_option, whole[0]):
            csets, general = whole[2], whole[3]
            steps_loop = ()
        else:
            steps_loop = steps
            plan["whole"] = None
        for k, (name, node, prim, ctx, d_x, d_e, d_r, L, pvals) in enumerate(steps_loop):
            cons = constraints_per_option[name]
            clist = cons.constraints if hasattr(cons, "constraints") else cons
            # A planner asks the same questions step after step: when an option's constraints are plain device-form dicts whose
            # every value is what it was at the last step (compared value by value: callers rewrite targets in place), the set of
            # the last step is the set of this one.  Anything else -- reference objects, a previous motion to align to -- takes the
            # general route (device form, structure and values keys, the shared cache).
            last = memo[k]
            if last is not None and prev_frames is None and skeleton is None and _same_constraints(clist, last[0]) and last[1].handle and \
                    last[1].cached_values is last[2] and getattr(cons, "hip_skeleton", None) is None and getattr(cons, "is_local", True):
                csets.append(last[1])      # (cached_values: nobody else rewrote the shared set)
                general.append(None)
            else:
                fp = flat_constraint_copy(clist) if prev_frames is None and skeleton is None else None
                sk = skeleton if skeleton is not None else getattr(cons, "hip_skeleton", None)
                form = constraints_to_device_form(clist)
                alignment = alignment_from_prev_frames(prev_frames, cons, sk)
                if any(is_frame_constraint(c) or c.get("type") == "trajectory" for c in form):
                    # trajectory and per-frame constraints are not keyframe channels: this option is scored by the general chain
                    # (device sampler, fused scorers, the per-frame kernels adding to the same errors, first minimum)
                    csets.append(None)
                    general.append((form, alignment, sk))
                    memo[k] = None
                    continue
                cs = cached_constraint_set(prim, form, sk, alignment)
                csets.append(cs)
                general.append((form, alignment, sk))
                memo[k] = (fp, cs, cs.cached_values) if fp is not None else None
        if steps_loop and prev_frames is None and skeleton is None and type(constraints_per_option) is dict and len(constraints_per_option) == len(steps) and \
                all(m is not None for m in memo) and all(cs is not None for cs in csets) and \
                all(getattr(constraints_per_option[st[0]], "hip_skeleton", None) is None for st in steps):
            plan["whole"] = ({st[0]: memo[k][0] for k, st in enumerate(steps)}, CSET_GENERATION[0], csets, general)
        on_device = bool(device_counts) and plan["one_context"] and steps and len(steps) <= 24 and all(cs is not None for cs in csets)
        if on_device:
            # the whole step on the device: counts (mg_options_counts_kernel), candidates, scores, first minima; records and
            # counts arrive in pinned host memory, one synchronisation
            m, stride, host = len(steps), plan["stride"], plan["host"]
            if steps_loop:
                for k, cs in enumerate(csets):
                    plan["csets"][k] = cs.handle.value
            np.add(plan["karange"], np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), out=plan["seeds_np"])
            rc = plan["lib"].mg_options_step_device_counts(m, plan["prims"], plan["csets"], n, plan["seeds"], plan["xs"], code, plan["lds"],
                                                           plan["errs"], plan["shared_ptr"], stride, plan["host_ptr"], plan["device_counts_ptr"])
            if rc == -4:           # MG_ERR_UNSUPPORTED: an option the one-launch kernel does not cover -- host draw below
                on_device = False
            else:
                if rc != 0:
                    _capi._check(rc)
                self._last_counts = (plan["device_counts"], steps)
                rec = host[:m * stride].copy().reshape(m, stride)
                errs = rec[:, 8:16].copy().view(np.float64)[:, 0].tolist()
                lat = rec[:, 16:].copy().view(np.float64)
                results = {st[0]: (lat[k, :st[7]], errs[k]) for k, st in enumerate(steps)}
                return options[int(np.argmin(errs))], results
        for k, st in enumerate(steps):   # the component counts, in option order, from NumPy's global stream
            plan["counts"][k, :len(st[8])] = multinomial(n, st[8])
        if any(cs is None for cs in csets) and plan["one_context"] and steps:
            mixed = self._mixed_step(plan, steps, csets, general, n, seed, dtype, code)
            if mixed is not None:
                return options[int(np.argmin([mixed[nm][1] for nm in options]))], mixed
        if any(cs is None for cs in csets):
            # at least one option needs the general chain and the mixed step does not cover it: the whole step goes option by option
            # (same draws: the sampler is keyed by seed + option index and the counts above)
            from .candidate_scoring import sample_rows_and_first_minimum
            results = {}
            for k, (name, node, prim, ctx, d_x, d_e, d_r, L, pvals) in enumerate(steps):
                form, alignment, sk = general[k] if general[k] is not None else (constraints_to_device_form(
                    constraints_per_option[name].constraints if hasattr(constraints_per_option[name], "constraints") else constraints_per_option[name]), None, None)
                idx, err, lat = sample_rows_and_first_minimum(node, form, alignment, plan["counts"][k, :len(pvals)].copy(), int(seed) + k, 0, n,
                                                              skeleton=sk, dtype=dtype)
                results[name] = (np.asarray(lat, dtype=np.float64), err)
            errors = [results[nm][1] for nm in options]
            return options[int(np.argmin(errors))], results
        results = {}
        if plan["one_context"] and steps:
            # one C call, ONE launch and ONE read-back for the whole step (mg_options_step): the result records side by side
            m, stride, host = len(steps), plan["stride"], plan["host"]
            if steps_loop:
                for k, cs in enumerate(csets):
                    plan["csets"][k] = cs.handle.value
            np.add(plan["karange"], np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), out=plan["seeds_np"])
            rc = plan["lib"].mg_options_step(m, plan["prims"], plan["csets"], n, plan["cnts"], plan["seeds"], plan["xs"], code, plan["lds"],
                                             plan["errs"], plan["shared_ptr"], stride, plan["host_ptr"])
            if rc != 0:
                _capi._check(rc)
            rec = host[:m * stride].copy().reshape(m, stride)          # one copy: the host block is reused by the next step
            errs = rec[:, 8:16].copy().view(np.float64)[:, 0].tolist()
            lat = rec[:, 16:].copy().view(np.float64)                    # (m, widest L): the winners, already rounded to the caller's type
            results = {st[0]: (lat[k, :st[7]], errs[k]) for k, st in enumerate(steps)}
            return options[int(np.argmin(errs))], results
        else:
            for k, (name, node, prim, ctx, d_x, d_e, d_r, L, pvals) in enumerate(steps):
                _capi._check(prim.lib.mg_option_step(prim.handle, csets[k].handle, n, plan["counts"][k].ctypes.data, int(seed) + k, d_x.ptr, code, L,
                                                     d_e.ptr, d_r.ptr))
            for name, node, prim, ctx, d_x, d_e, d_r, L, pvals in steps:
                raw = ctx.download(d_r, (16 + 8 * L,), np.uint8)       # synchronises this option's stream
                err = float(raw[8:16].view(np.float64)[0])
                results[name] = (raw[16:].view(np.float64).astype(dtype).astype(np.float64), err)
        errors = [results[n][1] for n in options]
        return options[int(np.argmin(errors))], results

    def _mixed_step(self, plan, steps, csets, general, n, seed, dtype, code):
        """A planner step in which some options carry trajectory or per-frame constraints (csets[k] is None for them): ONE launch
        still draws every option's candidates and scores their KEYFRAME constraints (mg_options_step, component counts already in
        the plan); the options with more then add the rest to their errors where the launch left them -- one launch per root
        trajectory, two per list of per-frame constraints (mg_joint_tracks + mg_score_frame_constraints) -- and take their own first
        minimum (one launch, one small read-back).  The additions are the general chain's, in its order: the same errors and
        winners, bit for bit (round 3 ran such a step option by option: sampler, scorer, ... per option).  None: not covered (an
        option without keyframe constraints, a trajectory aligned by another node than the root) -- the caller goes option by option."""
        from .candidate_scoring import cached_constraint_set, cached_trajectory, release_trajectory, split_trajectories
        from .frame_constraints import TrackScorer, split_frame_constraints, add_frame_constraints_dev
        plan["whole"] = None          # this step rewrites plan["csets"]: the whole-step shortcut must not score against them
        extras, sets = {}, list(csets)
        for k, st in enumerate(steps):
            if sets[k] is not None:
                continue
            form, alignment, sk = general[k]
            fused, frames = split_frame_constraints(form)
            keyframes, trajectories = split_trajectories(fused)
            if not keyframes:
                return None
            if alignment is not None and trajectories and alignment.get("joint", 0) not in (0, _capi.MG_ALIGN_START_POSE):
                return None
            sets[k] = cached_constraint_set(st[2], keyframes, sk, alignment)
            extras[k] = (trajectories, frames, alignment, sk)
        m, stride, host = len(steps), plan["stride"], plan["host"]
        for k, cs in enumerate(sets):
            plan["csets"][k] = cs.handle.value
        np.add(plan["karange"], np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), out=plan["seeds_np"])
        rc = plan["lib"].mg_options_step(m, plan["prims"], plan["csets"], n, plan["cnts"], plan["seeds"], plan["xs"], code, plan["lds"],
                                         plan["errs"], plan["shared_ptr"], stride, plan["host_ptr"])
        if rc == -4:      # MG_ERR_UNSUPPORTED: an option the one-launch kernel does not take
            return None
        if rc != 0:
            _capi._check(rc)
        rec = host[:m * stride].copy().reshape(m, stride)
        errs = rec[:, 8:16].copy().view(np.float64)[:, 0].tolist()
        lat = rec[:, 16:].copy().view(np.float64)
        results = {st[0]: (lat[k, :st[7]], errs[k]) for k, st in enumerate(steps) if k not in extras}
        scorers = plan.setdefault("track_scorers", {})
        # the options' trajectory constraints round by round (the j-th of every option that has one): ONE launch per round
        # (mg_score_trajectories) -- an option's own additions stay in its list's order
        for j in range(max(len(e[0]) for e in extras.values())):
            ks = [k for k, e in extras.items() if len(e[0]) > j]
            cj = [extras[k][0][j] for k in ks]
            # (pinned while the list is built and enqueued: more distinct trajectories than the cache holds must not close the first)
            trs = [cached_trajectory(steps[k][2], c, pin=True) for k, c in zip(ks, cj)]
            try:
                _capi.Primitive.score_trajectories_dev([steps[k][2] for k in ks], trs,
                                                       [steps[k][4] for k in ks], dtype, n, [steps[k][7] for k in ks], [steps[k][5] for k in ks],
                                                       [c.get("min_u", 0.0) for c in cj], [c.get("weight", 1.0) for c in cj],
                                                       [extras[k][2] for k in ks], accumulate=True)
            finally:
                for t in trs:
                    release_trajectory(t)
        # the options' per-frame lists and first minima: every option's joint tracks in ONE launch, every option's list + first minimum
        # in a second one, one read-back (mg_options_frame_lists; round 4: two launches per option + a first-minimum launch and two
        # synchronising reads per option).  An option whose list the call does not take (a joint-rotation constraint, more than four
        # constraints or requests) goes the per-option way below.
        fast, slow = [], []
        for k, (trajectories, frames, alignment, sk) in extras.items():
            scorer = None
            if frames:
                key = (k, _cs._freeze(frames), _cs._freeze(alignment), None if sk is None else sk.serial)
                scorer = scorers.get(key)
                if scorer and not scorer.valid():      # a cache was cleared under it
                    scorer.close()
                    scorer = None
                if scorer is None:
                    if len(scorers) > 64:
                        for old in scorers.values():
                            if old:
                                old.close()
                        scorers.clear()
                    try:
                        scorer = TrackScorer(steps[k][2], frames, sk, alignment)
                    except NotImplementedError:
                        scorer = False        # (a joint-rotation constraint, more than four requests: the frames chain)
                    scorers[key] = scorer
            if (not frames or (scorer and scorer.m <= _capi.MG_FRAME_LIST_MAX)) and plan["one_context"]:
                fast.append((k, scorer if frames else None))
            else:
                slow.append((k, scorer))
        if fast:
            out = _options_frame_lists(plan, steps, fast, extras, n, dtype)
            for (k, _), (idx, err, row) in zip(fast, out):
                results[steps[k][0]] = (row[:steps[k][7]].copy(), err)
        for k, scorer in slow:
            trajectories, frames, alignment, sk = extras[k]
            name, node, prim, ctx, d_x, d_e, d_r, L, pvals = steps[k]
            if scorer:
                scorer.score_dev(d_x, dtype, n, L, d_e, accumulate=True)
            else:
                add_frame_constraints_dev(prim, ctx.download(d_x, (n, L), dtype), frames, sk, alignment, d_e, accumulate=True)
            idx, err = ctx.argmin_first(d_e, n, np.float64)
            row = ctx.download(d_x.ptr.value + idx * L * np.dtype(dtype).itemsize, (L,), dtype)
            results[name] = (row.astype(np.float64), err)
        return results

    def options_step_rows(self, cmd, row_begin, row_end, skeleton=None):
        """One rank's share of a sharded planner step (distributed._cmd_options_step): the global rows [row_begin, row_end) of
        every option's draw, through mg_options_step_rows.  Returns {option: (global index, error, winning latent)}."""
        from .candidate_scoring import cached_constraint_set
        import ctypes as C
        options, n = tuple(cmd["options"]), int(cmd["n_samples"])
        dtype = np.dtype(cmd.get("dtype", "float32"))
        code = _capi.MG_F64 if dtype == np.float64 else _capi.MG_F32
        m_rows = int(row_end) - int(row_begin)
        plan = self._step_plan(options, m_rows, dtype)       # buffers sized for the block
        steps = plan["steps"]
        if not plan["one_context"]:
            raise NotImplementedError("sharded planner steps need all primitives in one context")
        m, stride, host = len(steps), plan["stride"], plan["host"]
        plan["whole"] = None          # plan["csets"] and the seeds are rewritten below
        for k, st in enumerate(steps):
            name = st[0]
            cs = cached_constraint_set(st[2], cmd["constraints"][name], skeleton, cmd["alignments"][name])
            plan["csets"][k] = cs.handle.value
            plan["seeds"][k] = int(cmd["seed"]) + k
            c = np.asarray(cmd["counts"][name], dtype=np.int64)
            plan["counts"][k, :len(c)] = c
        _capi._check(steps[0][2].lib.mg_options_step_rows(m, plan["prims"], plan["csets"], n, plan["cnts"], plan["seeds"], int(row_begin), m_rows,
                                                          plan["xs"], code, plan["lds"], plan["errs"], plan["shared"].ptr, stride,
                                                          host.ctypes.data_as(C.c_void_p)))
        out = {}
        for k, st in enumerate(steps):
            raw = host[k * stride:k * stride + 16 + 8 * st[7]]
            out[st[0]] = (int(raw[0:8].view(np.int64)[0]), float(raw[8:16].view(np.float64)[0]), raw[16:].view(np.float64).copy())
        return out

    def _step_plan(self, options, n, dtype):
        """Everything about a planner step that does not change from step to step, built once per (options, n, dtype): the
        per-option device buffers (no allocation inside a step), the argument arrays of mg_options_step, the normalised
        mixture weights the component counts are drawn with, the host block the result records land in."""
        import ctypes as C
        key = ("plan", options, n, dtype.str)
        plan = self._buffers.get(key)
        if plan is not None:
            return plan
        item = dtype.itemsize
        steps = []
        for name in options:
            node = self.nodes[name]
            prim, ctx = node._prim, node._prim.ctx
            L = prim.n_gmm_dims          # the winner comes back at full width (spatial + time latents)
            bkey = (name, n, dtype.str)
            bufs = self._buffers.get(bkey)
            if bufs is None:
                bufs = self._buffers[bkey] = (ctx.malloc(max(n, 1) * L * item), ctx.malloc(max(n, 1) * 8), ctx.malloc(16 + 8 * L))
            weights = np.asarray(node.gaussian_mixture_model.weights_, dtype=np.float64)
            steps.append((name, node, prim, ctx, bufs[0], bufs[1], bufs[2], L, weights / weights.sum()))
        m = len(steps)
        plan = {"steps": steps, "one_context": all(st[3] is steps[0][3] for st in steps),
                "counts": np.zeros((max(m, 1), max([len(st[8]) for st in steps] + [1])), dtype=np.int64)}
        if plan["one_context"] and steps:
            vp = C.c_void_p
            stride = 16 + 8 * max(st[7] for st in steps)
            plan.update(stride=stride, shared=steps[0][3].malloc(m * stride), host=np.empty(m * stride, dtype=np.uint8),
                        prims=(vp * m)(*[st[2].handle for st in steps]), csets=(vp * m)(),
                        cnts=(vp * m)(*[plan["counts"][k].ctypes.data for k in range(m)]), seeds=(C.c_uint64 * m)(),
                        xs=(vp * m)(*[_capi._dev_ptr(st[4]).value for st in steps]), lds=(C.c_int64 * m)(*[st[7] for st in steps]),
                        errs=(vp * m)(*[_capi._dev_ptr(st[5]).value for st in steps]), lib=steps[0][2].lib,
                        device_counts=np.zeros((m, 16), dtype=np.int64), karange=np.arange(m, dtype=np.uint64))
            # (everything a step hands to the C call, made once: no ctypes object is created inside a step)
            plan.update(seeds_np=np.ctypeslib.as_array(plan["seeds"]), shared_ptr=plan["shared"].ptr, host_ptr=plan["host"].ctypes.data_as(vp),
                        device_counts_ptr=plan["device_counts"].ctypes.data_as(vp))
        self._buffers[key] = plan
        return plan

    def evaluate_options(self, options, constraints_per_option, n_samples, rng_seed=None):
        """options: node names; constraints_per_option: name -> constraint list.  Returns
        (best_option, {name: (best_sample, min_error)})."""
        results = {}
        for name in options:
            node = self.nodes[name]
            prim = node._prim
            if rng_seed is not None:
                np.random.seed(rng_seed)
            samples = node.sample_low_dimensional_vector(n_samples)
            cset = _capi.ConstraintSet(prim, constraints_to_device_form(constraints_per_option[name]))
            try:
                S = _capi._latents(samples)
                d_s = self.ctx.upload(S)
                d_e = self.ctx.malloc(len(S) * 8)
                prim.score_constraints_dev(cset, d_s, S.dtype, len(S), S.shape[1], d_e, np.float64)
                idx, err = self.ctx.argmin_first(d_e, len(S), np.float64)
                d_s.free()
                d_e.free()
            finally:
                cset.close()
            results[name] = (samples[idx], err)
        errors = [results[n][1] for n in options]
        return options[int(np.argmin(errors))], results
