"""CPU oracle -- test infrastructure only (see oracle/hz_oracle.c header)."""
