"""PTLens lens model, Rd = a Ru^4 + b Ru^3 + c Ru^2 + (1 - a - b - c) Ru (reference corr_ca/model/ptlens.py)."""
import numpy as np

from .generic import NewtonRaphsonModel


class PtLensCorrectionModel(NewtonRaphsonModel):
    def __init__(self, a: float = 0, b: float = 0, c: float = 0):
        super().__init__()
        self._a = a
        self._b = b
        self._c = c

    def _undistorted_to_distorted(self, undistorted):
        sq = undistorted ** 2
        cube = undistorted * sq
        fourth = undistorted * cube
        return self._a * fourth + self._b * cube + self._c * sq + (1 - self._a - self._b - self._c) * undistorted

    def _undistorted_to_distorted_prior(self, undistorted):
        sq = undistorted ** 2
        cube = undistorted * sq
        return 4 * self._a * cube + 3 * self._b * sq + 2 * self._c * undistorted + (1 - self._a - self._b - self._c)

    def get_coefficients(self):
        return np.array((self._a, self._b, self._c))

    def compute_coefficients(self, r_distorted_undistorted):
        rd, ru = r_distorted_undistorted[:, 0], r_distorted_undistorted[:, 1]
        # Rd/Ru - 1 = a (Ru^3 - 1) + b (Ru^2 - 1) + c (Ru - 1), least squares
        basis = np.dstack((ru ** 3 - 1, ru ** 2 - 1, ru - 1))[0]
        try:
            self._a, self._b, self._c = np.linalg.lstsq(basis, (rd / ru) - 1)[0]
            return True
        except np.linalg.LinAlgError:
            return False
