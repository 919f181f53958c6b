"""Import-path twin of the reference's debayer/fast_resize.py: `debayer(image)` (fast_resize.py:7)."""
from . import debayer_fast as debayer  # noqa: F401
