"""sklearn-shaped Gaussian mixture backed by the HIP kernels.

The reference hands out a raw ``sklearn.mixture.GaussianMixture`` as
``MotionPrimitive.gaussian_mixture_model`` (reference
morphablegraphs/motion_model/motion_primitive.py:126-144) and callers use
``sample``, ``score``, ``score_samples`` and the fitted attributes
(``objective_functions.py:110,198-205``, ``time_constraints.py:95``).  This class
exposes the same names; ``score_samples`` runs ``mg_gmm_log_prob`` on the GPU.
"""
import numpy as np


def sample_like_sklearn(n_samples, weights, means, covars, random_state=None):
    """Bit-compatible restatement of ``GaussianMixture.sample`` for covariance_type='full'
    (what reference motion_primitive.py:189 calls): multinomial counts, then
    ``multivariate_normal`` per component, rows grouped by component and not shuffled,
    drawn from NumPy's global Mersenne state unless a RandomState is given."""
    rng = np.random.mtrand._rand if random_state is None else random_state
    if n_samples < 1:
        raise ValueError("Invalid value for 'n_samples': %d . The sampling requires at least one sample." % n_samples)
    counts = rng.multinomial(n_samples, weights)
    X = np.vstack([rng.multivariate_normal(mean, cov, int(c)) for mean, cov, c in zip(means, covars, counts)])
    y = np.concatenate([np.full(int(c), j, dtype=int) for j, c in enumerate(counts)])
    return X, y


class HipGaussianMixture(object):
    """Duck type of the fitted sklearn GaussianMixture the reference builds from JSON."""

    covariance_type = "full"
    converged_ = True

    def __init__(self, primitive, weights, means, covars):
        self._prim = primitive                       # _capi.Primitive
        self.weights_ = np.array(weights, dtype=np.float64)
        self.means_ = np.array(means, dtype=np.float64)
        self.covariances_ = np.array(covars, dtype=np.float64)
        self.n_components = len(self.weights_)
        self.n_dims = self.means_.shape[1]           # reference motion_primitive.py:143
        self._prec_chol = None

    @property
    def precisions_cholesky_(self):
        if self._prec_chol is None:
            self._prec_chol = self._prim.precisions_cholesky()
        return self._prec_chol

    def _check_X(self, X):
        X = np.asarray(X)
        if X.ndim != 2:
            # sklearn raises the same kind of error for the reference's 1-D call sites
            raise ValueError("Expected 2D array, got %dD array instead" % X.ndim)
        if X.shape[1] != self.n_dims:
            raise ValueError("X has %d features, but the mixture expects %d" % (X.shape[1], self.n_dims))
        return X

    def score_samples(self, X):
        """Per-sample log p(x), float64, computed by the HIP kernel."""
        return self._prim.gmm_log_prob(self._check_X(X), dtype=np.float64)

    def score(self, X, y=None):
        return float(self.score_samples(X).mean())

    def sample(self, n_samples=1, device=False, seed=None):
        """sklearn-compatible draw on the host RNG stream by default; ``device=True`` uses
        the Philox sampler on the GPU (same distribution, different stream)."""
        if not device:
            return sample_like_sklearn(n_samples, self.weights_, self.means_, self.covariances_)
        rng = np.random.mtrand._rand
        counts = rng.multinomial(n_samples, self.weights_)
        if seed is None:
            seed = int(rng.randint(0, 2 ** 31 - 1))
        X, comp = self._prim.gmm_sample(counts, seed, dtype=np.float64)
        return X, comp.astype(int)
