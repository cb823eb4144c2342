"""``Genie`` (reference genie/genie.py:18-181) -- out of scope for the hot path.

The reference class cannot be constructed at HEAD: ``__init__`` reads attributes it never defines and
``compute_loss`` passes a tuple where tokens are expected.  It is the "next" row 8f-4 of the
scope table; the three models it glues together (VideoTokenizer, LatentAction, DynamicsModel) are implemented here."""
from ._lightning import LightningModule


class Genie(LightningModule):
    def __init__(self, *args, **kwargs) -> None:
        raise NotImplementedError('genie.Genie is outside the implemented hot path (SURVEY.md section 8f-4): the reference class is '
                                  'unconstructible at HEAD; use VideoTokenizer, LatentAction and DynamicsModel directly')
