"""Lateral chromatic-aberration removal in the raw mosaic (reference corr_ca/): the apply half runs on the GPU,
`remove_ca_from_raw`; the lens models under `model/` evaluate their coordinate fields on the host.  Fitting a model
to an image (`compute_ca_lens_models_for_raw`, the structural-instability map and the tiled solver) is outside the
accelerated path and is not provided."""
from .ca_removal import remove_ca_from_raw  # noqa: F401
