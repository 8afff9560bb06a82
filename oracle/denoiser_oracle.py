"""CPU oracle of the denoiser stage.  TEST INFRASTRUCTURE ONLY -- never on the product path.

Restates reference wavjepa/denoiser.py on the building blocks of `oracle/jepa_oracle.py` (which are pinned to the reference
themselves):

  * forward / loss ............... denoiser.py:312-361   student encoder (conv -> LayerNorm -> mapper -> + positions -> post-norm ViT
                                                          -> final norm) on the clean clip and on the generated scene; targets =
                                                          `teacher.get_audio_representation(clean)` of a frozen JEPA;
                                                          loss = alpha * mse(clean) + (1 - alpha) * mse(generated)
  * batch hook ................... denoiser.py:217-309   scene generation (oracle/scene_oracle.py), 32 kHz -> sr resampling
                                                          (oracle/resample_oracle.py), shared crops + per-crop normalisation
                                                          (jepa_oracle.crop_normalize)

Parity pinning: `tests/test_oracle_golden.py::test_denoiser_oracle_matches_reference_fixture` checks the forward / loss / gradients
against `tests/golden/denoiser.npz`, produced by running the reference's own `Denoiser` (tests/golden/make_golden.py denoiser).
"""
from typing import Dict

import torch

from . import jepa_oracle as J


def contextual_features(P: Dict[str, torch.Tensor], audio: torch.Tensor, *, spec, enc_heads: int, mode: str) -> torch.Tensor:
    lf = J.local_features(P, audio, spec, mode)
    return J.encoder_stack(P, "encoder", lf, enc_heads, None, mode, final_norm=True)


def denoiser_forward(P, PT, generated: torch.Tensor, clean: torch.Tensor, *, alpha: float, spec, enc_heads: int, teacher_spec=None,
                     teacher_heads=None, mode: str = "fp32"):
    """P: student parameters (denoiser state_dict names), PT: the frozen JEPA's.  -> dict(loss, loss_clean, loss_denoise_dereverb)"""
    cf_clean = contextual_features(P, clean, spec=spec, enc_heads=enc_heads, mode=mode)
    cf_gen = contextual_features(P, generated, spec=spec, enc_heads=enc_heads, mode=mode)
    targets = J.audio_representation(PT, clean, None, spec=teacher_spec or spec, enc_heads=teacher_heads or enc_heads, mode=mode)
    lc = torch.nn.functional.mse_loss(cf_clean.float(), targets.float())
    lg = torch.nn.functional.mse_loss(cf_gen.float(), targets.float())
    return dict(loss=alpha * lc + (1.0 - alpha) * lg, loss_clean=lc, loss_denoise_dereverb=lg, targets=targets,
                contextual_features_clean=cf_clean, contextual_features_generated=cf_gen)
