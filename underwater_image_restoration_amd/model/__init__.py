"""Building blocks of the reference's ``src/model`` package (NewModel / NewBigFRFNModel family) on MI355X.
Round 2 holds ``block.FRFN`` (src/model/block.py:249-282); the blocks around it are SURVEY.md 8(f)-2, next."""
from .block import FRFN  # noqa: F401
