"""Import-path twin of the reference's Cython unit debayer/ahd_homogeneity_cython.pyx: `build_map` on the GPU."""
from .ahd_homogeneity import build_map  # noqa: F401
