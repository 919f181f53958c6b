from .cam_wb import CameraWhiteBalanceController  # noqa: F401
from .helpers_cam_mat import MatXyzToCamera, bradford_adapt_matrix  # noqa: F401
