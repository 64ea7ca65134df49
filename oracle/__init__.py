"""CPU oracle of the grid-erosion hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (soillib_amd/) must not.  Parity unpinned by the
reference — see oracle/soil_oracle.h.
"""
