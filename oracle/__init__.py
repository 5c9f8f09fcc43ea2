"""CPU oracle bindings (test infrastructure only).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``bwt_amd``) never does.
"""
from .binding import Oracle, Reference, load_oracle, load_reference  # noqa: F401
