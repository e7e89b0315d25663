"""Recipe configuration: the reference's hydra YAML layout read without hydra / omegaconf (neither exists on the GPU box).

The reference's trainings are launched as `fairseq-hydra-train --config-dir … --config-name transformer_ctc_librispeech
task.data=… task.dict=…` (examples/asr_librispeech/run_torchaudio.sh) with configs made of the groups `common, checkpoint,
task, dataset, distributed_training, criterion, optimization, optimizer, lr_scheduler, model, bpe`
(examples/asr_librispeech/config/*.yaml) over the defaults of fairseq/dataclass/configs.py.  This module gives the same file
the same meaning: group defaults (only the keys the path consumes), `${a.b.c}` interpolation, `???` as "must be given",
`group.key=value` command-line overrides with YAML scalar syntax, and the string-encoded tuples / lists the reference's
dataclasses `eval` (`adam_betas: (0.9,0.98)`, `conv_channels: "[64, 64, 128, 128]"`)."""
import ast
import copy
import re
from typing import Any, Dict, List, Optional

import yaml

# fairseq/dataclass/configs.py — defaults of the keys the training loop reads (CommonConfig :94-261, DatasetConfig :446-573,
# OptimizationConfig :576-640, CheckpointConfig :643-813, DistributedTrainingConfig :264-443)
DEFAULTS: Dict[str, Dict[str, Any]] = {
    "common": {"seed": 1, "log_interval": 100, "log_format": None, "empty_cache_freq": 0},
    "checkpoint": {
        "save_dir": "checkpoints", "restore_file": "checkpoint_last.pt", "reset_dataloader": False, "reset_lr_scheduler": False,
        "reset_meters": False, "reset_optimizer": False, "save_interval": 1, "save_interval_updates": 0,
        "keep_interval_updates": -1, "keep_last_epochs": -1, "keep_best_checkpoints": -1, "no_save": False,
        "no_epoch_checkpoints": False, "no_last_checkpoints": False, "no_save_optimizer_state": False,
        "best_checkpoint_metric": "loss", "maximize_best_checkpoint_metric": False, "patience": -1,
    },
    "dataset": {
        "max_tokens": None, "batch_size": None, "required_batch_size_multiple": 8, "train_subset": "train",
        "valid_subset": "valid", "validate_interval": 1, "validate_interval_updates": 0, "validate_after_updates": 0,
        "disable_validation": False, "max_tokens_valid": None, "batch_size_valid": None, "curriculum": 0, "num_workers": 1,
    },
    "distributed_training": {"distributed_world_size": 1, "ddp_backend": "legacy_ddp", "bucket_cap_mb": 25},
    "optimization": {
        "max_epoch": 0, "max_update": 0, "stop_time_hours": 0, "clip_norm": 0.0, "sentence_avg": False, "update_freq": [1],
        "lr": [0.25], "stop_min_lr": -1.0,
    },
    "optimizer": {"_name": "adam", "adam_betas": "(0.9, 0.999)", "adam_eps": 1e-8, "weight_decay": 0.0},
    "lr_scheduler": {"_name": "fixed"},
    "criterion": {"_name": "cross_entropy"},
    "task": {}, "model": {}, "bpe": {},
}

_INTERP = re.compile(r"\$\{([^}]+)\}")
# PyYAML follows YAML 1.1, where a float needs a dot: the recipes' `adam_eps: 1e-08`, `final_lr: 1e-6`, `warmup_init_lr: 1e-05`
# arrive as strings.  omegaconf (what the reference parses them with) reads them as floats — so does this loader.
_FLOAT = re.compile(r"[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+")


def _numbers(node):
    if isinstance(node, dict):
        return {k: _numbers(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_numbers(v) for v in node]
    if isinstance(node, str) and _FLOAT.fullmatch(node.strip()):
        return float(node)
    return node


def _lookup(cfg, dotted):
    cur = cfg
    for k in dotted.split("."):
        cur = cur[k]
    return cur


def _resolve(node, root):
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = _INTERP.fullmatch(node.strip())
        if m:  # whole-value interpolation keeps the referenced type (model_size: ${model.encoder.embed_dim} -> int)
            return _resolve(_lookup(root, m.group(1)), root)
        return _INTERP.sub(lambda mm: str(_resolve(_lookup(root, mm.group(1)), root)), node)
    return node


def _set(cfg, dotted, value):
    keys = dotted.split(".")
    cur = cfg
    for k in keys[:-1]:
        if not isinstance(cur.get(k), dict):
            cur[k] = {}
        cur = cur[k]
    cur[keys[-1]] = value


def _missing(node, prefix=""):
    out = []
    if isinstance(node, dict):
        for k, v in node.items():
            out += _missing(v, f"{prefix}{k}.")
    elif node == "???":
        out.append(prefix[:-1])
    return out


def load_config(path: Optional[str], overrides: Optional[List[str]] = None, base: Optional[dict] = None) -> dict:
    """YAML file (or `base` dict) + defaults + `a.b=value` overrides (hydra's `+a.b=value` / `++a.b=value` prefixes accepted),
    interpolations resolved; raises on any `???` left (hydra's MissingMandatoryValue)."""
    cfg = copy.deepcopy(DEFAULTS)
    user = yaml.safe_load(open(path)) if path else copy.deepcopy(base or {})
    for group, body in (user or {}).items():
        if isinstance(body, dict) and isinstance(cfg.get(group), dict):
            cfg[group].update(body)
        else:
            cfg[group] = body
    for ov in overrides or []:
        if "=" not in ov:
            raise ValueError(f"override must look like group.key=value, got {ov!r}")
        k, v = ov.split("=", 1)
        _set(cfg, k.lstrip("+"), yaml.safe_load(v) if v != "" else None)
    cfg = _numbers(_resolve(cfg, cfg))
    miss = _missing(cfg)
    if miss:
        raise ValueError("Missing mandatory value(s): " + ", ".join(miss))
    return cfg


def literal(v, default=None):
    """Values the reference's dataclasses keep as strings and `eval` (tuples, lists): `(0.9,0.98)` -> (0.9, 0.98)."""
    if v is None:
        return default
    if isinstance(v, str):
        return ast.literal_eval(v)
    return v


def as_list(v):
    v = literal(v)
    return list(v) if isinstance(v, (list, tuple)) else [v]


def per_epoch(values, epoch):
    """`update_freq` / `lr` lists are indexed by epoch and hold their last value (fairseq_cli/train.py:246-250)."""
    values = as_list(values)
    return values[min(epoch, len(values)) - 1]


# ---- legacy command lines (examples/asr_wsj/run.sh, examples/asr_swbd/run.sh: `fairseq_cli/train.py DATA --task … --arch … --flag value`) ----
# flag (underscored) -> configuration group; what is not listed belongs to the model (its --arch presets and overrides)
_LEGACY_GROUPS = {
    "common": ["seed", "log_interval", "log_format", "empty_cache_freq", "fp16", "amp", "tensorboard_logdir"],
    "dataset": ["num_workers", "data_buffer_size", "max_tokens", "batch_size", "curriculum", "valid_subset", "train_subset",
                "batch_size_valid", "max_tokens_valid", "required_batch_size_multiple", "validate_interval",
                "validate_interval_updates", "validate_after_updates", "disable_validation"],
    "distributed_training": ["ddp_backend", "distributed_world_size", "bucket_cap_mb"],
    "optimization": ["update_freq", "lr", "max_epoch", "max_update", "clip_norm", "sentence_avg", "stop_min_lr", "stop_time_hours"],
    "checkpoint": ["save_dir", "restore_file", "save_interval", "save_interval_updates", "keep_interval_updates", "keep_last_epochs",
                   "keep_best_checkpoints", "best_checkpoint_metric", "maximize_best_checkpoint_metric", "no_save",
                   "no_epoch_checkpoints", "no_last_checkpoints", "no_save_optimizer_state", "patience", "reset_optimizer",
                   "reset_lr_scheduler", "reset_dataloader", "reset_meters"],
    "optimizer": ["adam_betas", "adam_eps", "weight_decay"],
    "lr_scheduler": ["lr_shrink", "lr_threshold", "lr_patience", "warmup_updates", "warmup_init_lr", "start_reduce_lr_epoch",
                     "final_lr_scale", "warmup_steps", "hold_steps", "decay_steps", "phase_ratio", "init_lr_scale", "model_size",
                     "final_lr", "end_learning_rate", "power", "total_num_update"],
    "criterion": ["label_smoothing", "smoothing_type", "unigram_pseudo_count", "print_training_sample_interval", "zero_infinity"],
    "task": ["dict", "non_lang_syms", "word_dict", "wer_output_filter", "max_source_positions", "max_target_positions",
             "specaugment_config", "global_cmvn_stats_path", "feat_in_channels", "autoregressive", "include_eos_in_transducer_loss",
             "sample_break_mode", "tokens_per_sample", "output_dictionary_size", "is_wordlm"],
    "bpe": ["sentencepiece_model"],
}
_LEGACY_FLAG_GROUP = {f: g for g, fs in _LEGACY_GROUPS.items() for f in fs}
_LEGACY_SELECTORS = {"task": "task", "criterion": "criterion", "optimizer": "optimizer", "lr_scheduler": "lr_scheduler", "bpe": "bpe"}
_LIST_VALUED = {"update_freq", "lr", "scheduled_sampling_probs"}
# --arch -> registered model (fairseq/models/__init__.py ARCH_MODEL_REGISTRY of the reference's ASR architectures)
_ARCH_MODEL = {"speech_conv_lstm_wsj": "speech_lstm", "speech_conv_lstm_librispeech": "speech_lstm", "speech_conv_lstm_swbd": "speech_lstm",
               "speech_lstm": "speech_lstm", "lstm_lm_wsj": "lstm_lm_espresso", "lstm_lm_librispeech": "lstm_lm_espresso",
               "lstm_lm_swbd": "lstm_lm_espresso", "lstm_wordlm_wsj": "lstm_lm_espresso",
               "speech_transformer": "speech_transformer", "speech_transformer_wsj": "speech_transformer",
               "speech_transformer_librispeech": "speech_transformer", "speech_transformer_swbd": "speech_transformer"}


def is_legacy_argv(argv) -> bool:
    """fairseq_cli/train.py style (positional data directory and/or --task / --arch flags) rather than --config …"""
    return bool(argv) and not any(a in ("--config", "--config-dir", "--config-name") for a in argv) and (
        not argv[0].startswith("-") or "--task" in argv or "--arch" in argv)


def from_legacy_argv(argv: List[str]) -> dict:
    """The configuration a legacy command line describes, in the same grouped form `load_config` returns."""
    user: Dict[str, Dict[str, Any]] = {g: {} for g in DEFAULTS}
    i, positional = 0, []
    while i < len(argv):
        a = argv[i]
        if not a.startswith("--"):
            positional.append(a)
            i += 1
            continue
        key = a[2:].replace("-", "_")
        vals = []
        i += 1
        while i < len(argv) and not (argv[i].startswith("--") and not _FLOAT.fullmatch(argv[i])):
            vals.append(argv[i])
            i += 1
        parsed = [yaml.safe_load(v) for v in vals]
        value = True if not vals else (parsed if (len(parsed) > 1 or key in _LIST_VALUED) else parsed[0])
        if key in _LEGACY_SELECTORS:
            user[_LEGACY_SELECTORS[key]]["_name"] = value
        elif key == "arch":
            if value not in _ARCH_MODEL:
                raise NotImplementedError(f"--arch {value}: supported legacy architectures are {sorted(_ARCH_MODEL)} "
                                          "(the other Transformer / Conformer models are configured through the recipe YAMLs)")
            user["model"].update(_name=_ARCH_MODEL[value], arch=value)
        else:
            user[_LEGACY_FLAG_GROUP.get(key, "model")][key] = value
    if positional:
        user["task"]["data"] = positional[0]
    if "_name" not in user["task"]:
        user["task"]["_name"] = "speech_recognition_espresso"
    return load_config(None, [], base={g: b for g, b in user.items() if b})
