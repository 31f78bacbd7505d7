"""The sampling half of MotionPrimitiveGenerator
(reference morphablegraphs/motion_generator/motion_primitive_generator.py:47-261) on the HIP back end:
the three reference `constrained_sampling_mode`s plus the batched "gpu_batch" mode, the cluster-tree search
replaced by a brute-force pass over the tree's stored samples, and the local optimization
(reference optimization/least_squares.py:35-64) with the finite-difference Jacobian of the residual vector
evaluated in one launch per iteration instead of L + 1 objective calls.

Method names, the `algorithm_config` keys and the `data` tuples are the reference's.  Constraint construction and
graph walking stay in the reference.  With `use_local_coordinates` the constraints are localised first
(transform_constraints_to_local_cos) and no alignment is needed; without it the previous frames travel to the
scorer, which aligns every candidate to them on the device (candidate_scoring.alignment_from_prev_frames).
Anything the fused scorer does not cover raises NotImplementedError.
"""
import numpy as np
from scipy.optimize import leastsq

from . import objective_functions as of
from .candidate_scoring import SAMPLING_MODE_GPU_BATCH, evaluate_samples_using_constraints, sample_and_evaluate_on_device

SAMPLING_MODE_RANDOM = "random_discrete"                  # motion_primitive_generator.py:42-44
SAMPLING_MODE_CLUSTER_TREE_SEARCH = "cluster_tree_search"
SAMPLING_MODE_RANDOM_SPLINE = "random_spline"
SPATIAL_CONSTRAINT_TYPE_KEYFRAME_POSE = "keyframe_pose"   # constraints/spatial_constraints/__init__.py


class _EvaluationBudgetSpent(Exception):
    pass


class HipLeastSquares(object):
    """LeastSquares.run (reference optimization/least_squares.py:35-64): scipy's MINPACK Levenberg-Marquardt on
    the residual-vector objective.  MINPACK's own forward-difference Jacobian (lmdif / fdjac2: step sqrt(eps) * |x_j|,
    or sqrt(eps) where x_j == 0) is reproduced as `Dfun`, with its L objective evaluations batched into ONE launch.

    `max_iterations` keeps the reference's meaning.  The reference calls leastsq WITHOUT Dfun (least_squares.py:50-53),
    so MINPACK's lmdif charges the L evaluations of every finite-difference Jacobian against `maxfev`; with Dfun
    MINPACK (lmder) would count plain function calls only and the same number would buy about L + 1 times as many
    iterations.  The evaluations are therefore counted here the way lmdif counts them (1 per residual call, L per
    Jacobian) and the run stops where the reference's stops: when the count reaches max_iterations, with the best
    accepted point so far (Levenberg-Marquardt accepts a trial point exactly when it lowers the residual norm)."""

    def __init__(self, optimization_settings, objective=of.obj_spatial_error_residual_vector_and_naturalness):
        self.optimization_settings = optimization_settings
        self.verbose = optimization_settings.get("verbose", False)
        self._objective_function = objective
        self._error_func_params = None
        self.n_launches = 0
        self.n_equivalent_evaluations = 0     # what lmdif's nfev would read

    def set_objective_function(self, obj):
        self._objective_function = obj

    def set_objective_function_parameters(self, data):
        self._error_func_params = data

    def _func(self, s, data):
        key = np.asarray(s, dtype=np.float64).tobytes()
        if self._last_f is not None and self._last_f[0] == key:     # scipy checks the shapes with one call at x0, then MINPACK
            return self._last_f[1]                                   # asks for the same point again: one evaluation
        self.n_equivalent_evaluations += 1
        self.n_launches += 1
        r = np.asarray(self._objective_function(s, data), dtype=np.float64)
        self._last_f = (key, r)
        norm = float(np.dot(r, r))
        if norm < self._best[0]:                # an accepted Levenberg-Marquardt step
            self._best = (norm, np.array(s, dtype=np.float64))
        # lmdif tests nfev >= maxfev only after a TRIAL point has been evaluated and accepted or rejected (never after
        # the first residual call, never inside the Jacobian), so a run always completes the iteration it is in
        if self.n_equivalent_evaluations > 1 and self.n_equivalent_evaluations >= self._budget:
            raise _EvaluationBudgetSpent()
        return r

    def _jac(self, s, data):
        s = np.asarray(s, dtype=np.float64)
        key = s.tobytes()
        if self._last_j is not None and self._last_j[0] == key:
            return self._last_j[1]
        L = s.shape[0]
        self.n_equivalent_evaluations += L      # fdjac2 evaluates the residuals at L displaced points
        h = np.sqrt(np.finfo(np.float64).eps) * np.abs(s)
        h[h == 0.0] = np.sqrt(np.finfo(np.float64).eps)
        pts = np.repeat(s[None, :], L + 1, axis=0)
        pts[np.arange(1, L + 1), np.arange(L)] += h
        self.n_launches += 1
        r = self._objective_function(pts, data)             # (L + 1, m) in one launch
        J = ((r[1:] - r[:1]) / h[:, None]).T                 # (m, L)
        self._last_j = (key, J)
        return J

    def run(self, initial_guess):
        if self._objective_function is None or initial_guess is None:
            return initial_guess
        x0 = np.asarray(initial_guess, dtype=np.float64)
        self._budget = int(self.optimization_settings["max_iterations"])
        self.n_equivalent_evaluations = 0
        self._best = (np.inf, x0.copy())
        self._last_f = self._last_j = None
        try:
            result = leastsq(self._func, x0, args=(self._error_func_params,), Dfun=self._jac, maxfev=max(1, self._budget))
        except _EvaluationBudgetSpent:
            return self._best[1]
        except ValueError:
            return initial_guess
        return result[0]


class HipNumericalMinimizer(object):
    """NumericalMinimizer.run (reference optimization/numerical_minimizer.py:41-76): scipy.optimize.minimize on a scalar
    objective with the settings' method / tolerance / max_iterations / diff_eps.  Where the reference passes jac=None scipy
    differentiates the objective itself -- len(s) + 1 sequential objective calls per gradient, forward differences with the
    absolute step `eps`; here that same gradient ((f(s + eps e_i) - f(s)) / eps, scipy's '2-point' rule with abs_step = eps) is
    computed from ONE batched launch of the objective over the len(s) + 1 points, so the iterates are scipy's.  A Jacobian set
    with set_jacobian (step_goal_jac, obj_spatial_error_sum_and_naturalness_jac) is used as the reference uses it."""

    def __init__(self, optimization_settings, objective=None, jacobian=None):
        self.optimization_settings = optimization_settings
        self.verbose = optimization_settings.get("verbose", False)
        self._objective_function = objective
        self._jacobian = jacobian
        self._error_func_params = None
        self.n_launches = 0

    def set_objective_function(self, obj):
        self._objective_function = obj

    def set_objective_function_parameters(self, data):
        self._error_func_params = data

    def set_jacobian(self, jac):
        self._jacobian = jac

    def _fun(self, s, data):
        self.n_launches += 1
        return float(self._objective_function(np.asarray(s, dtype=np.float64), data))

    def _batched_forward_differences(self, s, data):
        s = np.asarray(s, dtype=np.float64)
        eps = float(self.optimization_settings.get("diff_eps", 1e-8) or 1e-8)
        L = s.shape[0]
        pts = np.repeat(s[None, :], L + 1, axis=0)
        pts[np.arange(1, L + 1), np.arange(L)] += eps
        self.n_launches += 1
        f = np.asarray(self._objective_function(pts, data), dtype=np.float64)       # (L + 1,) in one launch
        return (f[1:] - f[0]) / eps

    def run(self, initial_guess):
        if self._objective_function is None or initial_guess is None:
            return initial_guess
        from scipy.optimize import minimize
        st = self.optimization_settings
        jac = self._jacobian if self._jacobian is not None else self._batched_forward_differences
        derivative_free = str(st["method"]).lower() in ("nelder-mead", "powell", "cobyla")
        try:
            result = minimize(self._fun, np.asarray(initial_guess, dtype=np.float64), args=(self._error_func_params,), method=st["method"],
                              jac=None if derivative_free else (lambda s, data: np.asarray(jac(s, data), dtype=np.float64)),
                              tol=st.get("tolerance"), options={"maxiter": st["max_iterations"], "disp": bool(self.verbose)})
        except ValueError as e:              # the reference swallows it and returns the initial guess (numerical_minimizer.py:67-69)
            if self.verbose:
                print("Warning:", e.args)
            return initial_guess
        return result.x


class HipOptimizerBuilder(object):
    """OptimizerBuilder (reference optimization/optimizer_builder.py:36-88) over the batched objectives."""

    def __init__(self, algorithm_settings):
        self.algorithm_settings = algorithm_settings

    def build_spatial_and_naturalness_error_minimizer(self):
        st = self.algorithm_settings["local_optimization_settings"]
        if st.get("method", "leastsq") == "leastsq":
            return HipLeastSquares(st, of.obj_spatial_error_residual_vector_and_naturalness)
        return HipNumericalMinimizer(st, of.obj_spatial_error_sum_and_naturalness)

    def build_spatial_error_minimizer(self):
        st = self.algorithm_settings["local_optimization_settings"]
        if st.get("method", "leastsq") == "leastsq":
            return HipLeastSquares(st, of.obj_spatial_error_residual_vector)
        return HipNumericalMinimizer(st, of.obj_spatial_error_sum)

    def build_path_following_minimizer(self):
        return HipNumericalMinimizer(self.algorithm_settings["local_optimization_settings"], of.step_goal_error, of.step_goal_jac)

    def build_path_following_with_likelihood_minimizer(self):
        return HipNumericalMinimizer(self.algorithm_settings["local_optimization_settings"], of.step_goal_and_naturalness, of.step_goal_and_naturalness_jac)

    def build_time_error_minimizer(self):
        return HipNumericalMinimizer(self.algorithm_settings["global_time_optimization_settings"], of.obj_time_error_sum)

    def build_global_error_minimizer(self):
        return HipNumericalMinimizer(self.algorithm_settings["global_spatial_optimization_settings"], of.obj_global_error_sum)

    def build_global_error_minimizer_residual(self):
        return HipLeastSquares(self.algorithm_settings["global_spatial_optimization_settings"], of.obj_global_residual_vector_and_naturalness)


class HipMotionPrimitiveGenerator(object):
    """nodes: {(action_name, primitive_name): HipMotionStateGraphNode-like}; algorithm_config: the reference's
    dict (motion_generator/algorithm_configuration.py:30-110)."""

    def __init__(self, nodes, algorithm_config, action_name, prev_action_name=""):
        self.nodes = nodes
        self.action_name = action_name
        self.prev_action_name = prev_action_name
        self.set_algorithm_config(algorithm_config)
        self.numerical_minimizer = HipOptimizerBuilder(algorithm_config).build_spatial_and_naturalness_error_minimizer()
        self.objective = of.obj_spatial_error_sum

    def set_algorithm_config(self, algorithm_config):
        self._algorithm_config = algorithm_config
        self.n_random_samples = algorithm_config["n_random_samples"]
        self.use_constraints = algorithm_config.get("use_constraints", True)
        self._settings = algorithm_config["local_optimization_settings"]
        self.optimization_start_error_threshold = self._settings["start_error_threshold"]
        self.use_transition_model = algorithm_config.get("use_transition_model", False)
        self.constrained_sampling_mode = algorithm_config.get("constrained_sampling_mode", SAMPLING_MODE_GPU_BATCH)
        self.n_cluster_search_candidates = int(algorithm_config.get("n_cluster_search_candidates", 2))
        self.use_local_coordinates = algorithm_config.get("use_local_coordinates", True)
        # gpu_batch only: draw the candidates with the device sampler and keep them on the GPU (not sklearn's stream)
        # a distributed.MgCommunicator (or FileCommunicator) when the candidate loop is sharded over the GPUs of a node: this
        # process is rank 0, the other ranks run distributed.worker_loop over the same primitives
        self.communicator = algorithm_config.get("communicator", getattr(self, "communicator", None))
        self.gpu_sampling = bool(algorithm_config.get("gpu_sampling", False))
        self.gpu_sampling_seed = int(algorithm_config.get("gpu_sampling_seed", 0))

    # ---- motion_primitive_generator.py:78-124 ---------------------------------------------------------
    def generate_constrained_motion_spline(self, mp_constraints, prev_graph_walk=None):
        node_key = (self.action_name, mp_constraints.motion_primitive_name)
        steps = getattr(prev_graph_walk, "steps", [])
        if len(steps) > 0:
            prev_mp_name, prev_parameters = steps[-1].node_key[1], steps[-1].parameters
        else:
            prev_mp_name, prev_parameters = "", None
        if self.use_constraints and len(mp_constraints.constraints) > 0:
            prev_frames = prev_graph_walk.get_quat_frames() if len(steps) > 0 else None
            parameters = self.generate_constrained_sample(self.nodes[node_key], mp_constraints, prev_mp_name,
                                                          prev_frames, prev_parameters)
        else:
            parameters = self.generate_random_sample(node_key, prev_mp_name, prev_parameters)
        return self.nodes[node_key].back_project(parameters, use_time_parameters=False), parameters

    # ---- motion_primitive_generator.py:126-162 --------------------------------------------------------
    def generate_constrained_sample(self, graph_node, in_mp_constraints, prev_mp_name="", prev_frames=None,
                                    prev_parameters=None):
        if self.use_local_coordinates:
            prev_frames_copy = None
            if hasattr(in_mp_constraints, "transform_constraints_to_local_cos"):
                mp_constraints = in_mp_constraints.transform_constraints_to_local_cos()
            else:
                mp_constraints = in_mp_constraints
        else:
            mp_constraints = in_mp_constraints
            prev_frames_copy = prev_frames
        if self.constrained_sampling_mode == SAMPLING_MODE_RANDOM_SPLINE:
            raise NotImplementedError("random_spline scores through the proprietary mgrd package")
        elif self.constrained_sampling_mode == SAMPLING_MODE_CLUSTER_TREE_SEARCH and getattr(graph_node, "cluster_tree", None) is not None:
            sample = self._get_best_fit_sample_using_cluster_tree(graph_node, mp_constraints, prev_frames_copy)
        else:   # random_discrete and gpu_batch: draw n_random_samples, score all, first minimum
            sample = self._get_best_fit_sample_using_gmm(graph_node, mp_constraints, prev_mp_name, prev_frames_copy,
                                                         prev_parameters)
        if self._is_optimization_required(mp_constraints):
            sample = self._optimize_parameters_numerically(sample, graph_node, mp_constraints, prev_frames_copy)
        if mp_constraints is not in_mp_constraints:
            in_mp_constraints.min_error = mp_constraints.min_error
            in_mp_constraints.evaluations = mp_constraints.evaluations
        return sample

    def _is_optimization_required(self, mp_constraints):
        return getattr(mp_constraints, "use_local_optimization", False) and not self.use_transition_model and \
               mp_constraints.min_error >= self.optimization_start_error_threshold

    # ---- motion_primitive_generator.py:180-192 --------------------------------------------------------
    def _optimize_parameters_numerically(self, initial_guess, graph_node, mp_constraints, prev_frames):
        mp_constraints.constraints = [c for c in mp_constraints.constraints
                                      if getattr(c, "constraint_type", None) != SPATIAL_CONSTRAINT_TYPE_KEYFRAME_POSE]
        if len(mp_constraints.constraints) == 0:
            return initial_guess
        data = (graph_node, mp_constraints, prev_frames, self._settings["error_scale_factor"],
                self._settings["quality_scale_factor"], 1.0)
        error_sum = max(abs(np.sum(self.numerical_minimizer._objective_function(np.asarray(initial_guess), data))), 1.0)
        data = data[:5] + (error_sum,)
        self.numerical_minimizer.set_objective_function_parameters(data)
        return self.numerical_minimizer.run(initial_guess=initial_guess)

    # ---- motion_primitive_generator.py:194-209 --------------------------------------------------------
    def _get_best_fit_sample_using_gmm(self, graph_node, mp_constraints, prev_mp_name, prev_frames, prev_parameters):
        if self.use_transition_model and prev_parameters is not None:
            gmm = self._predict_gmm(mp_constraints.motion_primitive_name, prev_mp_name, prev_parameters)
            samples = gmm.sample(self.n_random_samples)
            samples = samples[0] if isinstance(samples, tuple) else samples
        elif self.gpu_sampling and self.constrained_sampling_mode == SAMPLING_MODE_GPU_BATCH:
            self.gpu_sampling_seed += 1
            best_sample, _ = sample_and_evaluate_on_device(graph_node, mp_constraints, self.n_random_samples,
                                                           self.gpu_sampling_seed, prev_frames=prev_frames, communicator=self.communicator)
            return best_sample
        else:
            samples = graph_node.sample_low_dimensional_vectors(self.n_random_samples)
        best_sample, _ = self.evaluate_samples_using_constraints(samples, graph_node, mp_constraints, prev_frames)
        return best_sample

    def generate_random_sample(self, node_key, prev_mp_name="", prev_parameters=None):
        prev = self.nodes.get((self.prev_action_name, prev_mp_name))
        if self.use_transition_model and prev_parameters is not None and prev is not None and prev.has_transition_model(node_key):
            return prev.predict_parameters(node_key, prev_parameters)
        return self.nodes[node_key].sample_low_dimensional_vector()

    def _predict_gmm(self, mp_name, prev_mp_name, prev_parameters):
        return self.nodes[(self.prev_action_name, prev_mp_name)].predict_gmm((self.action_name, mp_name), prev_parameters)

    # ---- motion_primitive_generator.py:220-228, cluster_tree.py:117-149 -------------------------------
    def _get_best_fit_sample_using_cluster_tree(self, graph_node, constraints, prev_frames, n_candidates=-1):
        """The reference descends the k-means / KD tree with `n_candidates` kept per level, calling the objective
        once per visited mean or sample.  At GPU batch sizes the whole tree is cheaper to score than to descend:
        every stored sample in one launch, first minimum -- the result of find_best_example_exhaustive."""
        stored = np.asarray(graph_node.cluster_tree.data)[:, :graph_node.get_n_spatial_components()]
        best, distance = evaluate_samples_using_constraints(stored, graph_node, constraints, prev_frames, communicator=self.communicator)
        constraints.min_error = distance
        return np.array(best)

    # ---- motion_primitive_generator.py:230-261 --------------------------------------------------------
    def evaluate_samples_using_constraints(self, samples, mp_node, constraints, prev_frames):
        return evaluate_samples_using_constraints(samples, mp_node, constraints, prev_frames, communicator=self.communicator)
