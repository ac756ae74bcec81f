"""Drop-in for ``utils.contrastive_utils.contrastive_loss`` (the PCA visualisers of the reference
module are cosmetic and out of scope).  NOTE: placing this directory first on PYTHONPATH shadows the
reference's whole ``utils`` package; INTEGRATION.md shows the one-line import patch instead."""
from instascene_amd.contrastive import contrastive_loss  # noqa: F401
