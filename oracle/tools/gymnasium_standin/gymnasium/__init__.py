"""Minimal stand-in for the `gymnasium` package (absent from this image).

Test infrastructure only: it lets `oracle/tools/gen_goldens.py` import the
read-only reference checkout in the build container to emit golden vectors.
It is never imported by the product package and never travels as a dependency.
The surface is exactly what the reference touches: `gymnasium.Env` as a base
class and `gymnasium.spaces.Box(low, high, shape, dtype)` with `.shape`,
`.sample()`, `.contains()`.
"""
from . import spaces  # noqa: F401


class Env:
    def __init__(self, *args, **kwargs):
        pass
