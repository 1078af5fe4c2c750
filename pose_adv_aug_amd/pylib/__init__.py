"""Device-backed counterparts of the reference's pylib/ package (same module and function names)."""
from . import HumanPts, HumanAug, Evaluation, HumanAcc, Criterion  # noqa: F401
