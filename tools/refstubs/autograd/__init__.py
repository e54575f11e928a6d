"""Forward-only stand-in for HIPS autograd (absent from this image). No AD."""
