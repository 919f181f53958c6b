# The reference's colorize/__init__.py is empty although its README imports from the package
# (README.md:56); both spellings work here.
from .transform import (cam_to_clean_xyz, cam_to_lin_srgb, cam_to_rgb_norm, clip_rgb, final_matrix,  # noqa: F401
                        lin_srgb_to_srgb, srgb_to_lin_srgb)
