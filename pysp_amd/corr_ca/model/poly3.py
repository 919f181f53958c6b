"""Poly3 lens model, Rd = k1 Ru^3 + (1 - k1) Ru (reference corr_ca/model/poly3.py)."""
import numpy as np

from .generic import NewtonRaphsonModel


class Poly3CorrectionModel(NewtonRaphsonModel):
    def __init__(self, initial_k1: float = 0):
        self._k1 = min(1.0, max(initial_k1, 0.0))                        # poly3.py:23: k1 is kept inside [0, 1]
        super().__init__()

    def _undistorted_to_distorted(self, undistorted):
        return self._k1 * undistorted ** 3 + (1 - self._k1) * undistorted

    def _undistorted_to_distorted_prior(self, undistorted):
        return 3 * self._k1 * undistorted ** 2 + (1 - self._k1)

    def get_coefficients(self):
        return np.array((self._k1))

    def compute_coefficients(self, r_distorted_undistorted: np.ndarray):
        rd, ru = r_distorted_undistorted[:, 0], r_distorted_undistorted[:, 1]
        self._k1 = np.median(((rd / ru) - 1) / (ru ** 2 - 1))            # Rd/Ru - 1 = k1 (Ru^2 - 1)
        return True
