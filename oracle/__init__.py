"""CPU oracle (test infrastructure only; see gcnn_oracle.py).  PARITY UNPINNED: the reference ships no golden
vectors and its TensorFlow runtime is absent here."""
