#!/usr/bin/env python3
"""Denoiser-stage training entry point for MI355X -- the surface of the reference's denoise.py (same config tree `configs/denoise.yaml`
+ groups, same component wiring: extractor -> Denoiser -> WebAudioDataModuleDenoiser -> trainer with gradient_clip_val 1.0; the student
starts from the pre-trained WavJEPA checkpoint, which is also the frozen teacher) on the HIP engine.

    python denoise.py trainer.teacher_ckpt_weights=runs/last.ckpt data.data_dir=/corpus/a-{000..099}.tar data.rir_dir=... data.noise_dir=...
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 denoise.py trainer.num_gpus=8 ...
"""
import os
import sys

import torch

from utils import checkpoint_dir, get_identity_from_cfg_denoise
from wavjepa_amd.config import load_config, parse_conv_spec
from wavjepa_amd.data_modules import WebAudioDataModuleDenoiser
from wavjepa_amd.denoiser import Denoiser
from wavjepa_amd.extractors import ConvFeatureExtractor, Extractor
from wavjepa_amd.trainer import Trainer
from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG

EXTRACTORS = {"wav2vec2": ConvFeatureExtractor, "wavjepa": ConvFeatureExtractor}


class ComponentFactory:
    @staticmethod
    def create_extractor(cfg) -> Extractor:
        cls = EXTRACTORS.get(cfg.extractor.name)
        if cls is None:
            raise ValueError(f"Unknown extractor type: {cfg.extractor.name}. Available extractors: {list(EXTRACTORS.keys())}")
        return cls(conv_layers_spec=parse_conv_spec(cfg.extractor.conv_layers_spec), in_channels=cfg.data.in_channels,
                   depthwise=cfg.extractor.depthwise)

    @staticmethod
    def create_network(cfg, extractor: Extractor) -> Denoiser:
        return Denoiser(feature_extractor=extractor, transformer_encoder_cfg=TransformerEncoderCFG.create(),
                        transformer_encoder_layers_cfg=TransformerLayerCFG.create(), lr=cfg.optimizer.lr,
                        adam_betas=(cfg.optimizer.b1, cfg.optimizer.b2), adam_weight_decay=cfg.optimizer.weight_decay,
                        resample_sr=cfg.data.sr, process_audio_seconds=cfg.data.process_seconds,
                        nr_samples_per_audio=cfg.data.samples_per_audio, size=cfg.trainer.get("size", "base"), alpha=cfg.trainer.alpha)


def setup_trainer(cfg) -> Trainer:
    return Trainer(accelerator=cfg.trainer.accelerator, max_epochs=cfg.trainer.epochs, max_steps=cfg.trainer.steps,
                   precision=cfg.trainer.precision, devices=int(cfg.trainer.num_gpus), gradient_clip_val=1.0, gradient_clip_algorithm="norm",
                   log_every_n_steps=cfg.trainer.get("log_every_n_steps", 1),
                   default_root_dir=checkpoint_dir(cfg, "saved_models_jepa_denoised", get_identity_from_cfg_denoise(cfg)),
                   checkpoint_every_n_steps=2500)


def create_data_module(cfg, nr_patches, rank: int):
    return WebAudioDataModuleDenoiser(data_dir=cfg.data.data_dir, noise_dir=cfg.data.noise_dir, rir_dir=cfg.data.rir_dir,
                                      batch_size=cfg.trainer.batch_size, nr_samples_per_audio=cfg.data.samples_per_audio,
                                      nr_time_points=nr_patches, with_rir=cfg.data.with_rir, with_noise=cfg.data.with_noise,
                                      snr_high=cfg.data.snr_high, snr_low=cfg.data.snr_low, seed=cfg.seed, rank=rank,
                                      world_size=int(cfg.trainer.num_gpus))


def build_model(cfg):
    extractor = ComponentFactory.create_extractor(cfg)
    return ComponentFactory.create_network(cfg, extractor), extractor.total_patches(int(cfg.data.sr * cfg.data.process_seconds))


def main(argv=None):
    cfg = load_config(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs"), list(argv if argv is not None else sys.argv[1:]),
                      config_name="denoise")
    try:
        torch.manual_seed(cfg.seed)
        trainer = setup_trainer(cfg)
        model, patches = build_model(cfg)
        data_module = create_data_module(cfg, patches, trainer.rank)
        if trainer.rank == 0:
            print(f"Effective Batch Size is: {cfg.trainer.batch_size * cfg.data.samples_per_audio * cfg.trainer.num_gpus}")
        weights = torch.load(cfg.trainer.teacher_ckpt_weights, weights_only=False)
        state = {k.replace("._orig_mod", ""): v for k, v in weights["state_dict"].items()}
        mine = model.state_dict()
        model.load_state_dict({k: v for k, v in state.items() if k in mine and tuple(v.shape) == tuple(mine[k].shape)}, strict=False)
        model._set_teacher(cfg.trainer.teacher_ckpt_weights)
        trainer.fit(model, data_module, ckpt_path=cfg.get("ckpt_path", None))
    except Exception as e:
        print(f"Training failed with error: {str(e)}")
        raise


if __name__ == "__main__":
    main()
