"""Building blocks of the reference's ``src/model`` package (NewModel / NewBigFRFNModel family) on MI355X.
Round 2 holds ``block.FRFN`` / ``block.LeFF`` / ``block.FDFP`` (src/model/block.py:223-282, 518-556) and ``model.EncoderBlock`` in its Fourier mode
(src/model/model.py:16-93); the decoder half (DecoderBlock, MDASSA with its window attention) is SURVEY.md 8(f)-2, next."""
from .block import FRFN, LeFF, FDFP  # noqa: F401
from .model import DecoderBlock, EncoderBlock, MyBigFRFNModel  # noqa: F401
