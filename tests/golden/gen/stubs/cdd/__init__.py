"""Empty stand-in for pycddlib (import-only; golden-vector generation)."""
