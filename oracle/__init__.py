"""CPU oracle: test infrastructure only.  See oracle/cube_oracle.py for the rules on who may import it."""
