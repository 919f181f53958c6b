"""Poly5 lens model, Rd = Ru + h1 Ru^3 + h2 Ru^5 (reference corr_ca/model/poly5.py)."""
import numpy as np

from .generic import NewtonRaphsonModel


class Poly5CorrectionModel(NewtonRaphsonModel):
    def __init__(self, h1: float = 0, h2: float = 0):
        super().__init__()
        self._h1 = h1
        self._h2 = h2

    def _undistorted_to_distorted(self, undistorted):
        sq = undistorted ** 2
        cube = undistorted * sq
        fifth = cube * sq
        return undistorted + self._h1 * cube + self._h2 * fifth

    def _undistorted_to_distorted_prior(self, undistorted):
        sq = undistorted ** 2
        fourth = sq * sq
        return 5 * self._h2 * fourth + 3 * self._h1 * sq + 1

    def get_coefficients(self):
        return np.array((self._h1, self._h2))

    def compute_coefficients(self, r_distorted_undistorted):
        rd, ru = r_distorted_undistorted[:, 0], r_distorted_undistorted[:, 1]
        basis = np.dstack((ru ** 3, ru ** 5))[0]                         # Rd - Ru = h1 Ru^3 + h2 Ru^5, least squares
        try:
            self._h1, self._h2 = np.linalg.lstsq(basis, rd - ru)[0]
            return True
        except np.linalg.LinAlgError:
            return False
