"""Few-shot methods on the hot path (mirror of the reference's ``methods`` package)."""
from . import meta_template, gnn, gnnnet, gnnnet_copy, baselinefinetune  # noqa: F401
