"""oracle -- CPU restatement of the reference's damaged-pixel detection path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (image_processing_amd/,
image_processing.py) never does.  See oracle/oracle.c for provenance and the
parity status of each function.
"""
