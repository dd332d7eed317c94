"""CPU oracle (test infrastructure).  See maxsim_oracle.py's header for the import rules."""
