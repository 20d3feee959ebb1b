"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's bundle-adjustment hot path plus the harness that runs the real
reference (when /root/reference exists) through import shims.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import from here; the product (multical_amd/) never does.
"""
