"""Version of the reference API this package mirrors (PyChebyshev v0.21.1)."""
__version__ = "0.21.1"
__backend__ = "hip-gfx950"
