"""Radial lens models for chromatic-aberration alignment (reference corr_ca/model/generic.py).

A model maps an undistorted normalised radius to a distorted one.  Its coordinate field only ever needs the top-left
quadrant of the frame: the other three are mirror images (generic.py:84-99), which is also the form the GPU consumes
(`*_quadrant`, (H/2, W/2, 2) float32 holding (dy, dx) from the image centre).  The arithmetic below keeps the
reference's dtypes and operation order -- including the float64 detour that appears when the coefficients are NumPy
float64 scalars, as they are after a fit -- so the fields are bit-identical to the reference's.
"""
from __future__ import annotations

from abc import abstractmethod

import numpy as np

from ... import _hostpar


def _even_shape(image: np.ndarray):
    rows, cols = image.shape[:2]
    if rows % 2 or cols % 2:
        raise ValueError("Incorrect shape for packing!")               # generic.py:8-11
    return rows, cols


def get_empty_coord_field(image: np.ndarray) -> np.ndarray:
    """(H/2, W/2, 2) int32 grid of (row, column) indices of the top-left quadrant (generic.py:6-18)."""
    rows, cols = _even_shape(image)
    grid = np.empty((rows // 2, cols // 2, 2), dtype=np.int32)
    grid[..., 0] = np.arange(rows // 2, dtype=np.int32)[:, None]
    grid[..., 1] = np.arange(cols // 2, dtype=np.int32)[None, :]
    return grid


def get_empty_radius_field(image: np.ndarray) -> np.ndarray:
    """Distance of every top-left-quadrant pixel from the image centre, 1.0 at the corner (generic.py:20-41)."""
    rows, cols = _even_shape(image)
    r = np.zeros((rows // 2, cols // 2), dtype=np.float32)
    r[:, ] = (np.arange(cols // 2)[::-1] + 0.5) ** 2                     # float64 squares stored as float32
    r += ((np.arange(rows // 2)[::-1] + 0.5) ** 2)[:, np.newaxis]      # float64 add, rounded back to float32
    r = np.sqrt(r)
    return r / r[0, 0]


def mirror_quadrant(quad: np.ndarray, shape) -> np.ndarray:
    """Full (H, W, 2) field from its top-left quadrant: columns mirror with dx negated, rows mirror with dy negated."""
    h, w = quad.shape[:2]
    out = np.zeros((shape[0], shape[1], 2), dtype=np.float32)
    out[:h, :w] = quad
    out[:h, w:, 0] = quad[:, ::-1, 0]
    out[:h, w:, 1] = -quad[:, ::-1, 1]
    out[h:, :, 0] = -out[:h][::-1, :, 0]
    out[h:, :, 1] = out[:h][::-1, :, 1]
    return out


def _radius_rows(rows: int, cols: int, sl: slice) -> np.ndarray:
    """Rows `sl` of get_empty_radius_field before the final division (same expressions, row by row independent)."""
    r = np.zeros((sl.stop - sl.start, cols // 2), dtype=np.float32)
    r[:, ] = (np.arange(cols // 2)[::-1] + 0.5) ** 2
    r += ((np.arange(rows // 2)[::-1] + 0.5) ** 2)[sl, np.newaxis]
    return np.sqrt(r)


def _quadrant(image: np.ndarray, radial) -> np.ndarray:
    """Offsets from the centre scaled by radial(r) / r (generic.py:68-82 and :137-149).

    Evaluated in row blocks on the host team (pysp_amd/_hostpar.py): every pass is elementwise, so the blocks hold the bits of the
    whole-array expressions above; `radial` sees the whole flattened radius field, as in the reference (the Newton inversion's stopping
    rule is global), and threads its own passes."""
    rows, cols = _even_shape(image)
    h, w = rows // 2, cols // 2
    parts = _hostpar.blocks(h, _hostpar.team() if h * w >= _hostpar.MIN_PARALLEL_ELEMS else 1)
    if len(parts) == 1:
        radius = get_empty_radius_field(image)
    else:
        radius = np.concatenate(_hostpar.pmap(lambda sl: _radius_rows(rows, cols, sl), parts))
        r00 = radius[0, 0]
        _hostpar.pmap(lambda sl: np.divide(radius[sl], r00, out=radius[sl]), parts)
    centre = (np.array(image.shape[:2]) - 1) / 2
    mapped = radial(radius.reshape(-1)).reshape(-1, radius.shape[1])
    off = np.empty((h, w, 2), dtype=np.float32)

    def finish(sl: slice) -> None:
        o = np.empty((sl.stop - sl.start, w, 2), dtype=np.int32)
        o[..., 0] = np.arange(sl.start, sl.stop, dtype=np.int32)[:, None]
        o[..., 1] = np.arange(w, dtype=np.int32)[None, :]
        o = o.astype(np.float32)
        o[..., 0] -= centre[0]
        o[..., 1] -= centre[1]
        gain = mapped[sl] / radius[sl]
        o[..., 0] *= gain
        o[..., 1] *= gain
        off[sl] = o

    _hostpar.pmap(finish, parts)
    return off


def _elementwise(fn, x: np.ndarray) -> np.ndarray:
    """fn(x) for an elementwise fn, evaluated in blocks of the flattened array on the host team."""
    flat = x.reshape(-1)
    if flat.size < _hostpar.MIN_PARALLEL_ELEMS or _hostpar.team() == 1:
        return fn(x)
    parts = _hostpar.blocks(flat.size, _hostpar.team())
    return np.concatenate(_hostpar.pmap(lambda sl: fn(flat[sl]), parts)).reshape(x.shape)


class CaCorrectionModel:
    @abstractmethod
    def compute_coefficients(self, r_distorted_undistorted: np.ndarray) -> bool:
        ...

    @abstractmethod
    def get_coefficients(self) -> np.ndarray:
        ...

    @abstractmethod
    def get_distorted(self, undistorted: np.ndarray) -> np.ndarray:
        ...

    def compute_error_statistics(self, r_distorted_undistorted: np.ndarray):
        raise NotImplementedError("")

    def get_distorted_quadrant(self, image: np.ndarray) -> np.ndarray:
        return _quadrant(image, self.get_distorted)

    def get_distorted_coordinates(self, image: np.ndarray) -> np.ndarray:
        """Where each undistorted position lands under the model; cv2.remap with it undoes the distortion."""
        return mirror_quadrant(self.get_distorted_quadrant(image), image.shape[:2])


class ReversibleModelMixin:
    @abstractmethod
    def estimate_undistorted(self, distorted: np.ndarray, max_iterations: int = 8, max_epsilon: float = 0.00001) -> np.ndarray:
        ...

    def get_undistorted_quadrant(self, image: np.ndarray) -> np.ndarray:
        return _quadrant(image, self.estimate_undistorted)

    def get_undistorted_coordinates(self, image: np.ndarray) -> np.ndarray:
        """The inverse field: where each distorted position came from."""
        return mirror_quadrant(self.get_undistorted_quadrant(image), image.shape[:2])


class NewtonRaphsonModel(CaCorrectionModel, ReversibleModelMixin):
    """Polynomial models inverted with Newton's method on g(Ru) = f(Ru) - Rd (generic.py:165-203)."""

    @abstractmethod
    def _undistorted_to_distorted(self, undistorted: np.ndarray) -> np.ndarray:
        ...

    @abstractmethod
    def _undistorted_to_distorted_prior(self, undistorted: np.ndarray) -> np.ndarray:
        ...

    def get_distorted(self, undistorted):
        if isinstance(undistorted, np.ndarray):
            return _elementwise(self._undistorted_to_distorted, undistorted)
        return self._undistorted_to_distorted(undistorted)

    def estimate_undistorted(self, distorted: np.ndarray, max_iterations: int = 8, max_epsilon: float = 0.00001) -> np.ndarray:
        # The stopping rule is global (largest change over the whole field), so the number of steps -- and with it
        # every value -- depends on the frame size; it is evaluated here exactly as the reference does.
        n_parts = _hostpar.team() if isinstance(distorted, np.ndarray) and distorted.size >= _hostpar.MIN_PARALLEL_ELEMS else 1
        if n_parts == 1:
            estimate = np.zeros_like(distorted)
            previous_step = np.inf
            for _ in range(max_iterations):
                before = np.copy(estimate)
                estimate = estimate - ((self._undistorted_to_distorted(estimate) - distorted) / self._undistorted_to_distorted_prior(estimate))
                step = np.max(np.abs(before - estimate))
                if step < max_epsilon or step == previous_step:
                    break
                previous_step = step
            return estimate
        # the same iteration on blocks of the flattened field, one thread each; the blocks meet after every step for the global maximum
        # (each block's estimate is re-bound like the whole array's, so a float64 coefficient widens it at the same step)
        flat = distorted.reshape(-1)
        parts = _hostpar.blocks(flat.size, n_parts)
        est = [np.zeros_like(flat[sl]) for sl in parts]

        def newton(i: int):
            d, before = flat[parts[i]], est[i]
            est[i] = before - ((self._undistorted_to_distorted(before) - d) / self._undistorted_to_distorted_prior(before))
            return np.max(np.abs(before - est[i]))

        previous_step = np.inf
        for _ in range(max_iterations):
            steps = _hostpar.pmap(newton, range(len(parts)))
            step = steps[0]
            for s_ in steps[1:]:
                step = s_ if s_ > step or s_ != s_ else step            # np.max propagates NaN
            if step < max_epsilon or step == previous_step:
                break
            previous_step = step
        return np.concatenate(est).reshape(distorted.shape)
