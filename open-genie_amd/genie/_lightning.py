"""Optional Lightning: the reference's models are ``LightningModule``s (reference genie/tokenizer.py:13,225).
When ``lightning`` is installed the real base classes are used; otherwise a minimal stand-in keeps the same
methods callable so the hot path and ``genie/trainer.py`` work without it."""
import torch.nn as nn

try:                                                   # pragma: no cover - depends on the environment
    from lightning import LightningDataModule, LightningModule
    HAVE_LIGHTNING = True
except Exception:                                      # lightning absent (this image): inert stand-ins
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *args, **kwargs) -> None:
            self.hparams = dict(kwargs)

        def log_dict(self, metrics, *args, **kwargs) -> None:
            self._last_logged = dict(metrics)

        def log(self, name, value, *args, **kwargs) -> None:
            self._last_logged = {name: value}

    class LightningDataModule:
        def __init__(self, *args, **kwargs) -> None:
            pass

        def save_hyperparameters(self, *args, **kwargs) -> None:
            pass
