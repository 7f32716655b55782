"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/pvamd_oracle.c).  Importable from tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke(); never from pytorch_volumetric_amd/."""
