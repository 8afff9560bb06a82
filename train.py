#!/usr/bin/env python3
"""WavJEPA pre-training on MI355X: same wiring as the reference's train.py (registries, ComponentFactory, trainer
setup: reference train.py:21-35,47-130,160-206,225-250) on top of the HIP engine.

    python train.py                                   # configs/base.yaml
    python train.py masker=LibriSpeech trainer.steps=1000 trainer.batch_size=32
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py trainer.num_gpus=8
"""
from __future__ import annotations

import os
import sys

import torch

from utils import checkpoint_dir, get_identity_from_cfg
from wavjepa_amd.config import load_config, parse_conv_spec
from wavjepa_amd.data import SyntheticAudioSource
from wavjepa_amd.extractors import ConvChannelFeatureExtractor, ConvFeatureExtractor, Extractor
from wavjepa_amd.jepa import JEPA
from wavjepa_amd.masking import SpeechMasker, TimeInverseBlockMasker
from wavjepa_amd.trainer import Trainer
from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG

NETWORKS = {"JEPA": JEPA}
MASKERS = {"time-inverse": TimeInverseBlockMasker, "speech-masker": SpeechMasker}
EXTRACTORS = {"wav2vec2": ConvFeatureExtractor, "wavjepa": ConvFeatureExtractor,
              # BASELINE config 4 (WavJEPA-Nat): the reference's registry (train.py:26-29) never selects its channel extractor; this one does
              "wavjepa-nat": ConvChannelFeatureExtractor}


class ComponentFactory:
    @staticmethod
    def create_extractor(cfg) -> Extractor:
        cls = EXTRACTORS.get(cfg.extractor.name)
        if cls is None:
            raise ValueError(f"Unknown extractor type: {cfg.extractor.name}. Available extractors: {list(EXTRACTORS.keys())}")
        if cls is ConvChannelFeatureExtractor:
            return cls(conv_layers_spec=parse_conv_spec(cfg.extractor.conv_layers_spec), in_channels=cfg.data.in_channels,
                       share_weights_over_channels=bool(cfg.extractor.get("share_weights_over_channels", False)))
        return cls(conv_layers_spec=parse_conv_spec(cfg.extractor.conv_layers_spec), in_channels=cfg.data.in_channels,
                   depthwise=cfg.extractor.depthwise)

    @staticmethod
    def create_masker(cfg):
        cls = MASKERS.get(cfg.masker.name)
        if cls is None:
            raise ValueError(f"Unknown masker type: {cfg.masker.name}. Available maskers: {list(MASKERS.keys())}")
        mk = cfg.masker
        if mk.name == "speech-masker":
            return SpeechMasker(target_masks_per_context=mk.target_masks_per_context, target_prob=mk.target_prob,
                                target_length=mk.target_length, ratio_cutoff=mk.ratio_cutoff,
                                channel_based_masking=mk.channel_based_masking, min_context_len=mk.min_context_len)
        # the reference reads cfg.masker.context_prob / context_length while its YAML spells context_mask_prob /
        # context_mask_length (SURVEY §5 drift 1): accept both spellings.
        prob = mk.get("context_prob", mk.get("context_mask_prob"))
        length = mk.get("context_length", mk.get("context_mask_length"))
        return TimeInverseBlockMasker(target_masks_per_context=mk.target_masks_per_context, context_mask_prob=prob,
                                      context_mask_length=length, target_prob=mk.target_prob, target_length=mk.target_length,
                                      ratio_cutoff=mk.ratio_cutoff, channel_based_masking=mk.channel_based_masking,
                                      channel_major=bool(mk.get("channel_major", False)))

    @staticmethod
    def create_network(cfg, extractor: Extractor) -> JEPA:
        cls = NETWORKS.get(cfg.model)
        if cls is None:
            raise ValueError(f"Unknown network type: {cfg.model}. Available networks: {list(NETWORKS.keys())}")
        try:
            return cls(feature_extractor=extractor, transformer_encoder_cfg=TransformerEncoderCFG.create(),
                       transformer_encoder_layers_cfg=TransformerLayerCFG.create(), transformer_decoder_cfg=TransformerEncoderCFG.create(),
                       transformer_decoder_layers_cfg=TransformerLayerCFG.create(d_model=384), lr=cfg.optimizer.lr,
                       adam_betas=(cfg.optimizer.b1, cfg.optimizer.b2), adam_weight_decay=cfg.optimizer.weight_decay,
                       resample_sr=cfg.data.sr, process_audio_seconds=cfg.data.process_seconds,
                       nr_samples_per_audio=cfg.data.samples_per_audio, compile_modules=cfg.trainer.compile_modules,
                       average_top_k_layers=cfg.trainer.average_top_k_layers, size=cfg.trainer.get("size", "base"),
                       warmup_steps=cfg.trainer.get("warmup_steps", 100000))
        except Exception as e:
            raise RuntimeError(f"Failed to create network instance: {str(e)}")


def setup_trainer(cfg) -> Trainer:
    return Trainer(accelerator=cfg.trainer.accelerator, max_epochs=cfg.trainer.epochs, max_steps=cfg.trainer.steps,
                   precision=cfg.trainer.precision, devices=int(cfg.trainer.num_gpus), gradient_clip_val=5,
                   gradient_clip_algorithm="norm", strategy="ddp" if int(cfg.trainer.num_gpus) > 1 else "auto",
                   log_every_n_steps=cfg.trainer.get("log_every_n_steps", 50),
                   checkpoint_every_n_steps=25000,      # ModelCheckpoint(every_n_train_steps=25000, save_last=True), reference train.py:146-153
                   default_root_dir=checkpoint_dir(cfg, "saved_models_jepa_new_masking", get_identity_from_cfg(cfg)))


def build_model(cfg):
    extractor = ComponentFactory.create_extractor(cfg)
    network = ComponentFactory.create_network(cfg, extractor)
    return network, extractor.total_patches(int(cfg.data.sr * cfg.data.process_seconds))


def create_data_source(cfg, nr_patches, device, rank):
    masker = ComponentFactory.create_masker(cfg)
    if cfg.data.name == "NatSynthetic":     # BASELINE config 4: binaural scenes generated on the device inside the step
        from wavjepa_amd.data import NatSceneSource
        if int(cfg.data.in_channels) != 2:
            raise ValueError("data=nat_synthetic produces 2-channel scenes: data.in_channels must be 2")
        return NatSceneSource(masker, batch_size=cfg.trainer.batch_size, samples_per_audio=cfg.data.samples_per_audio, n_tokens=nr_patches,
                              sr=cfg.data.sr, seconds=cfg.data.get("source_seconds", 10.0), rir_seconds=cfg.data.get("rir_seconds", 0.5),
                              n_noise=int(cfg.data.get("n_noise", 2)), seed=cfg.seed + rank, device=device)
    if cfg.data.name != "Synthetic":        # tar shards of .flac clips (reference train.py:94-110 -> data_modules/WebAudioDataModule.py)
        from wavjepa_amd.data_modules import WebAudioDataModule
        dm = WebAudioDataModule(masker, data_dirs=cfg.data.data_dirs, mixing_weights=cfg.data.get("mixing_weights", None),
                                batch_size=cfg.trainer.batch_size, nr_samples_per_audio=cfg.data.samples_per_audio,
                                nr_time_points=nr_patches, in_channels=cfg.data.in_channels, sr=cfg.data.sr, seed=cfg.seed, rank=rank,
                                world_size=cfg.trainer.num_gpus)
        dm.setup("fit")
        return dm.train_dataloader()
    return SyntheticAudioSource(masker, batch_size=cfg.trainer.batch_size, samples_per_audio=cfg.data.samples_per_audio,
                                n_tokens=nr_patches, in_channels=cfg.data.in_channels, sr=cfg.data.sr,
                                seconds=cfg.data.get("source_seconds", 10.0), seed=cfg.seed + rank, device=device)


def main(argv=None):
    cfg = load_config(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs"), list(argv if argv is not None else sys.argv[1:]))
    try:
        torch.manual_seed(cfg.seed)
        trainer = setup_trainer(cfg)
        model, patches = build_model(cfg)
        device = torch.device("cuda", trainer.local_rank)
        source = create_data_source(cfg, patches, device, trainer.rank)
        if trainer.rank == 0:
            print(f"Effective Batch Size is: {cfg.trainer.batch_size * cfg.data.samples_per_audio * cfg.trainer.num_gpus}")
        trainer.fit(model, train_dataloaders=source)
    except Exception as e:
        print(f"Training failed with error: {str(e)}")
        raise


if __name__ == "__main__":
    main()
