"""Stub-import recipe for the read-only reference (test tooling; runs ONLY in the build container).

The reference (`/root/reference`, labhamlet/wavjepa) imports pytorch_lightning, torchaudio and
webdataset at package import time; none is installed here.  We register minimal stand-in modules
in ``sys.modules`` *for the import only* (they provide no numerics: every number the reference
produces still comes from its own code on top of stock torch), then import ``wavjepa.jepa``.

Nothing in this file travels to the GPU box as a dependency: `-m gpu` tests, smoke() and bench.py
never import it (the reference directory does not exist there).
"""
import sys
import types
import inspect

REFERENCE_ROOT = "/root/reference"


class _AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def install_stubs():
    import transformers  # noqa: F401  (must be imported before the torchaudio stub exists)
    import torch
    from torch import nn

    if "pytorch_lightning" in sys.modules and getattr(sys.modules["pytorch_lightning"], "_is_stub", False):
        return

    pl = types.ModuleType("pytorch_lightning")
    pl._is_stub = True

    class _Trainer:
        max_steps = 1000

    class LightningModule(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.global_step = 0
            self.trainer = _Trainer()
            self._hp = _AttrDict()

        def save_hyperparameters(self, ignore=()):
            frame = inspect.currentframe().f_back
            loc = frame.f_locals
            for k, v in loc.items():
                if k in ("self", "__class__", "kwargs") or k in ignore:
                    continue
                self._hp[k] = v

        @property
        def hparams(self):
            return self._hp

        @property
        def device(self):
            return next(self.parameters()).device

        def log_dict(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.LightningDataModule = LightningDataModule
    pl.Trainer = _Trainer
    pl.seed_everything = lambda s, workers=False: torch.manual_seed(s)
    sys.modules["pytorch_lightning"] = pl
    for sub in ("callbacks", "loggers"):
        m = types.ModuleType(f"pytorch_lightning.{sub}")
        for name in ("LearningRateMonitor", "ModelCheckpoint", "TensorBoardLogger"):
            setattr(m, name, object)
        sys.modules[f"pytorch_lightning.{sub}"] = m

    ta = types.ModuleType("torchaudio")
    ta.functional = types.ModuleType("torchaudio.functional")
    ta.transforms = types.ModuleType("torchaudio.transforms")
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.functional"] = ta.functional
    sys.modules["torchaudio.transforms"] = ta.transforms

    wds = types.ModuleType("webdataset")
    wds.RandomMix = object
    wds.WebDataset = object
    wds.warn_and_continue = None
    wds.split_by_node = None
    wds.split_by_worker = None
    sys.modules["webdataset"] = wds

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def import_reference():
    install_stubs()
    import wavjepa.jepa as ref_jepa
    import wavjepa.masking as ref_masking
    import wavjepa.audio_masking as ref_audio_masking
    import wavjepa.pos_embed as ref_pos_embed
    from wavjepa.extractors import ConvFeatureExtractor
    from wavjepa.types import TransformerEncoderCFG, TransformerLayerCFG
    return types.SimpleNamespace(
        jepa=ref_jepa, masking=ref_masking, audio_masking=ref_audio_masking,
        pos_embed=ref_pos_embed, ConvFeatureExtractor=ConvFeatureExtractor,
        TransformerEncoderCFG=TransformerEncoderCFG, TransformerLayerCFG=TransformerLayerCFG)
