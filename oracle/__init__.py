"""CPU oracle (test infrastructure only -- see oracle/desenet_ref.py header)."""
