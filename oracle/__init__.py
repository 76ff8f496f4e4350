"""CPU oracle package (TEST INFRASTRUCTURE ONLY -- see oracle/sgw_oracle.h)."""
