"""CPU oracle — test infrastructure only (see oracle/molar_oracle.h)."""
