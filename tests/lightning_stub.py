"""A minimal stand-in for pytorch_lightning (which this image does not ship), used ONLY by
tests/test_host_cpu.py::test_jepa_is_a_lightning_module_when_lightning_is_importable to check that wavjepa_amd.JEPA derives
from LightningModule when Lightning is importable.  It reproduces the properties of the real class that matter here:
`global_step`, `device` and `hparams` are READ-ONLY properties, `trainer` raises while no Trainer is attached, and
`save_hyperparameters` captures the constructor arguments of the calling frame."""
import inspect
import sys
import types

import torch
from torch import nn


class _AttributeDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class LightningModule(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        self._trainer = None
        self._hparams = _AttributeDict()
        self.logged = []

    @property
    def hparams(self):
        return self._hparams

    @property
    def global_step(self):
        return self._trainer.global_step if self._trainer is not None else 0

    @property
    def trainer(self):
        if self._trainer is None:
            raise RuntimeError(f"{type(self).__name__} is not attached to a `Trainer`.")
        return self._trainer

    @trainer.setter
    def trainer(self, t):
        self._trainer = t

    @property
    def device(self):
        return next(self.parameters()).device

    def save_hyperparameters(self, *args, ignore=None, **kw):
        frame = inspect.currentframe().f_back
        names = inspect.getfullargspec(type(self).__init__).args[1:]
        for n in names:
            if n in frame.f_locals and n not in (ignore or []):
                self._hparams[n] = frame.f_locals[n]
        for k, v in frame.f_locals.get("kwargs", {}).items():
            self._hparams[k] = v

    def log_dict(self, data, **kw):
        self.logged.append(dict(data))


def install():
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = LightningModule
    pl.__version__ = "stub"
    sys.modules["pytorch_lightning"] = pl
    return pl
