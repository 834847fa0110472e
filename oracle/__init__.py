"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference's algorithms for the hot path.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; nothing
under cream_amd/ does (tests/test_cabi.py::test_product_does_not_import_oracle checks).
"""
