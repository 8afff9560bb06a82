"""Run identity strings: where checkpoints (and, upstream, TensorBoard logs) of a configuration go (reference utils.py:1-43).

The fields, their order, the getters and the fall-back values are the reference's, so that a run started there and a run started
here with the same configuration share a directory: `{save_dir}/saved_models_jepa_new_masking/<identity with '_' -> '/'>`
(train.py:144-148) and `{save_dir}/saved_models_jepa_denoised/<...>` (denoise.py:92-96).  Upstream reads `context_prob` /
`context_len` while its YAML spells `context_mask_prob` / `context_mask_length` (drift 1): the names are kept as
read, so the fall-backs 0.65 / 10 appear in the string exactly as they do there."""


def _join(fields) -> str:
    return "_".join(f"{label}={value}" for label, value in fields)


def _common(cfg):
    return [("Data", cfg.data.get("name", None)), ("Extractor", cfg.extractor.name), ("InSeconds", cfg.data.process_seconds),
            ("BatchSize", cfg.trainer.get("batch_size")), ("NrSamples", cfg.data.get("samples_per_audio")),
            ("NrGPUs", cfg.trainer.get("num_gpus")), ("LR", cfg.optimizer.get("lr"))]


def get_identity_from_cfg(cfg) -> str:
    mk = cfg.masker
    return _join(_common(cfg) + [("TargetProb", mk.get("target_prob", 0.25)), ("TargetLen", mk.get("target_length", 10)),
                                 ("ContextProb", mk.get("context_prob", 0.65)), ("ContextLen", mk.get("context_len", 10)),
                                 ("MinContextBlock", mk.get("min_context_len", 1)), ("ContextRatio", mk.get("ratio_cutoff", 0.1))])


def get_identity_from_cfg_denoise(cfg) -> str:
    return _join(_common(cfg) + [("Alpha", cfg.trainer.get("alpha", 0.0))])


def checkpoint_dir(cfg, stage_dir: str, identity: str) -> str:
    return f"{cfg.save_dir}/{stage_dir}/{identity.replace('_', '/')}"
