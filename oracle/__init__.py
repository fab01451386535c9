"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's algorithm for the MedPLIB hot path (torch fp32 on the host).  Importable only from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product package `medplib_amd` never imports it.
Pinning status per module is stated in each module's docstring and in DESIGN.md."""
