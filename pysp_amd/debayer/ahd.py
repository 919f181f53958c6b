"""Import-path twin of the reference's debayer/ahd.py: `debayer(image, postprocess_stages=1)` (ahd.py:14)."""
from . import debayer_ahd as debayer  # noqa: F401
