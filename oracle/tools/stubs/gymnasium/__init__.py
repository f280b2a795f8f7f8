"""Minimal stand-in for `gymnasium` used ONLY by oracle/tools/make_golden.py in the
build container so that the read-only Python reference can be imported to emit
golden vectors.  Not part of the product, never imported by it."""


class Env:
    pass


from . import spaces  # noqa: E402,F401
