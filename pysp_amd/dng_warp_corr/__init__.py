from .chan_distortion_corr import apply_opcode_3_warp, stack_warp_prior  # noqa: F401
from .dng_warp_rectilinear_coords import compute_offset_remapping_table, compute_remapping_table  # noqa: F401
