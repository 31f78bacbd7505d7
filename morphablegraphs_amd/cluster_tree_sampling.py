"""The data-producing half of ClusterTreeBuilder (reference morphablegraphs/construction/cluster_tree_builder.py:
159-192, 293-301) on the HIP back end: drawing the training samples of a space partitioning, filtering them by
likelihood, and back-projecting them to frames for the feature maps.  The reference scores and back-projects one
sample at a time (`gmm.score([s])[0]`, `spline.evaluate(frame_idx)` per frame); here each method is one batched
call.  The Euclidean feature map of the feature trees (features.py:125-153: global positions of a set of joints in
every frame of every sample, then sklearn's PCA keeping 95 % of the variance) gets its positions from one forward-kinematics
launch (mg_joint_positions).  Clustering itself (k-means / KD tree construction, sklearn) is offline tooling and stays
where it is.

Method names and results are the reference's; `motion_primitive` is a HipMotionPrimitiveModelWrapper or a
HipMotionStateGraphNode.  There is no CPU fallback.
"""
import heapq

import numpy as np


def _log_likelihoods(motion_primitive, samples):
    """`get_gaussian_mixture_model().score([s])[0]` for every row (the reference's mixture wrappers return
    per-sample log-likelihoods from score(); extended_mgrd_mixture_model.py:100-108), in one launch."""
    return np.asarray(motion_primitive.get_gaussian_mixture_model().score_samples(np.asarray(samples)), dtype=np.float64)


class HipClusterTreeSampler(object):
    def __init__(self, n_samples=10000):
        self.n_samples = int(n_samples)          # cluster_tree_builder.py:124, config "n_random_samples"

    def set_config(self, config):
        self.n_samples = int(config["n_random_samples"])

    # cluster_tree_builder.py:159-175
    def _get_samples_using_threshold(self, motion_primitive, threshold=0, max_iter_count=5):
        """Rounds of n_samples draws, keeping the rows whose log-likelihood exceeds `threshold`, until n_samples
        are kept (every row of the round that crosses the line is kept, like the reference) or the rounds run
        out -> None."""
        data = []
        count = 0
        iter_count = 0
        while count < self.n_samples and iter_count < max_iter_count:
            samples = np.asarray(motion_primitive.sample_low_dimensional_vectors(self.n_samples))
            keep = _log_likelihoods(motion_primitive, samples) > threshold
            data.extend(samples[keep])
            count += int(keep.sum())
            iter_count += 1
        if iter_count < max_iter_count:
            return np.asarray(data)
        return None

    # cluster_tree_builder.py:177-189
    def _get_best_samples(self, motion_primitive):
        """2 n draws, likelihoods pushed on a heap as (-likelihood, idx); the FIRST n entries of the heap's list
        (heap order, not sorted order -- kept as the reference has it), shuffled."""
        data = np.asarray(motion_primitive.sample_low_dimensional_vectors(self.n_samples * 2))
        likelihoods = []
        for idx, likelihood in enumerate(_log_likelihoods(motion_primitive, data)):
            heapq.heappush(likelihoods, (-float(likelihood), idx))
        indices = [idx for _, idx in likelihoods[:self.n_samples]]
        data = data[indices]
        np.random.shuffle(data)
        return data

    # cluster_tree_builder.py:191-192
    def sample_data(self, motion_primitive):
        return motion_primitive.sample_low_dimensional_vectors(self.n_samples)

    # cluster_tree_builder.py:293-301
    def _back_project(self, mp, data):
        """(n, F, D): every sample evaluated at the integer canonical frames 0 .. F-1 (`spline.evaluate(frame_idx)`),
        float64 through the float64 frames path."""
        prim_obj = mp.motion_primitive if hasattr(mp, "motion_primitive") else mp
        prim = prim_obj._prim
        n_frames = int(mp.get_n_canonical_frames())
        S = np.asarray(data, dtype=np.float64)[:, :prim.n_components]
        grid = prim.time_grid(np.arange(n_frames, dtype=np.float64))
        try:
            return prim.back_project_frames_f64(S, grid)
        finally:
            grid.close()

    # cluster_tree_builder.py:266-291
    def _extract_features(self, motion_primitive, data, feature_type="latent", skeleton=None, joint_names=None, step=1):
        """The features a FeatureClusterTree is built on: the spatial latents themselves, or (feature_type "euclidean_pca",
        the reference's FEATURE_TYPE_EUCLIDEAN_PCA) the PCA projection of every sample's joint point cloud over its frames.
        skeleton: a _capi.Skeleton; joint_names: the reference passes END_EFFECTORS2 and step = 1."""
        if feature_type != "euclidean_pca":
            n_spatial = motion_primitive.get_n_spatial_components()
            return np.asarray(data)[:, :n_spatial]
        motions = self._back_project(motion_primitive, data)
        return self.map_motions_to_euclidean_pca(motion_primitive, motions, skeleton, joint_names, step)[0]

    # space_partitioning/features.py:133-153
    def map_motions_to_euclidean_space(self, motion_primitive, motions, skeleton, joint_names, step=4):
        """(n, F * len(joint_names) * 3): per sample the global positions of `joint_names` in its frames, flattened.  As
        in the reference a frame whose index is not a multiple of `step` REPEATS the cloud of the last one that is (its
        `sample.append(frame)` sits outside the `if idx % step == 0`), so the vector always has F entries."""
        prim_obj = motion_primitive.motion_primitive if hasattr(motion_primitive, "motion_primitive") else motion_primitive
        ctx = prim_obj._prim.ctx
        motions = np.asarray(motions, dtype=np.float64)
        n, F, D = motions.shape
        keep = np.arange(0, F, int(step))
        pos = ctx.joint_positions(skeleton, joint_names, motions[:, keep, :].reshape(-1, D)).reshape(n, len(keep), -1)
        return pos[:, np.arange(F) // int(step), :].reshape(n, -1)

    # space_partitioning/features.py:125-130
    def map_motions_to_euclidean_pca(self, motion_primitive, motions, skeleton, joint_names, step=4):
        from sklearn.decomposition import PCA
        point_clouds = self.map_motions_to_euclidean_space(motion_primitive, motions, skeleton, joint_names, step)
        pca = PCA(n_components=0.95)
        return pca.fit_transform(point_clouds), pca
