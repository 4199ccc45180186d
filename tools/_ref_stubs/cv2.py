"""Inert stand-in for OpenCV, used ONLY by tools/gen_golden.py in the build container.
models/agc.py imports cv2 at module scope (agc.py:3) but the live hot path never calls it
(its only use is visualisation, agc.py:291)."""
