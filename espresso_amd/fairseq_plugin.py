"""Reference-side binding: espresso_amd's task / models / criterions behind fairseq's own registries and call contracts.

`install()` (called by `registry.mirror_into_fairseq()`) needs fairseq + espresso importable (the training host of a user who
switches; NOT the GPU test box, which has neither — there the repo's own `speech_train.py` / `speech_recognize.py` drive the
same classes).  After it,

    fairseq.tasks.setup_task(cfg.task)            -> FairseqTaskAdapter (a FairseqTask) around SpeechRecognitionEspressoTask
    task.build_model(cfg.model)                   -> fairseq.models.build_model -> MODEL / ARCH registries -> espresso_amd model
    task.build_criterion(cfg.criterion)           -> fairseq.criterions.build_criterion -> espresso_amd criterion
    task.train_step / valid_step / optimizer_step / inference_step / build_generator / load_dataset / get_batch_iterator ...

follow the signatures of fairseq/tasks/fairseq_task.py:121,132,207,327,344,359,490,524,530,538 and resolve to this package.
The reference's config dataclasses stay registered (fairseq merges the user's config into them, fairseq/tasks/__init__.py:36-39,
fairseq/models/__init__.py:83-90); `adapt_cfg` then narrows the merged config to the fields the HIP path implements and REFUSES
a non-default value of a field it does not implement instead of ignoring it.

Pinned by tests/test_fairseq_binding.py (CPU, reference imported through oracle/ref_stubs): resolution, signatures, model /
criterion construction.  Not exercised: a full fairseq-CLI run on the MI355X (fairseq is not installed on the GPU box)."""
import dataclasses
import inspect

import torch

from . import registry


def _plain(x):
    """OmegaConf node / dataclass instance / Namespace -> plain nested python."""
    if dataclasses.is_dataclass(x) and not isinstance(x, type):
        return {f.name: _plain(getattr(x, f.name)) for f in dataclasses.fields(x)}
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if hasattr(x, "__dict__") and type(x).__name__ == "Namespace":
        return {k: _plain(v) for k, v in vars(x).items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_plain(v) for v in x) if not hasattr(x, "_fields") else x
    if hasattr(x, "value") and x.__class__.__module__.startswith("fairseq.dataclass"):  # ChoiceEnum members
        return x.value
    return x


def adapt_cfg(cfg, target_cls, ref_defaults=None, where=""):
    """Merged reference-style config -> instance of espresso_amd's config dataclass `target_cls`.  A field the target does not have
    must be None, an unresolved `${...}` interpolation, or equal to the reference dataclass's own default (`ref_defaults`):
    anything else is a feature the HIP path would silently drop, so it raises."""
    d = _plain(cfg)
    d.pop("_name", None)
    ref_defaults = ref_defaults or {}
    known = {f.name: f for f in dataclasses.fields(target_cls)}
    kw, bad = {}, {}
    for k, v in d.items():
        if k in known:
            sub = known[k].default_factory if known[k].default_factory is not dataclasses.MISSING else None
            if isinstance(sub, type) and dataclasses.is_dataclass(sub) and isinstance(v, dict):
                kw[k] = adapt_cfg(v, sub, ref_defaults.get(k) if isinstance(ref_defaults.get(k), dict) else None, f"{where}{k}.")
            else:
                kw[k] = v
        elif v is not None and not (isinstance(v, str) and v.startswith("${")) and (k not in ref_defaults or ref_defaults[k] != v):
            bad[k] = v  # (also a key the reference's own dataclass does not have: nothing says it is safe to ignore)
    if bad:
        raise NotImplementedError(f"espresso_amd does not implement {where}{sorted(bad)} (non-default values {bad})")
    return target_cls(**kw)


def _model_entry(name, cls, owner=None):
    """What fairseq.models.build_model calls `.build_model(cfg, task)` on: adapts the merged config, builds OUR model.
    `owner`: the registered model an architecture alias belongs to — the reference's dataclass (whose defaults decide which
    foreign keys may be ignored) is registered under the MODEL's name, not under the alias."""
    cfg_cls = getattr(cls, "config_class", None)

    class Entry(cls):  # a subclass so that isinstance / registry introspection still see the espresso_amd model
        @classmethod
        def build_model(kls, cfg, task):
            inner_task = getattr(task, "inner", task)
            if cfg_cls is not None and not isinstance(cfg, cfg_cls):
                import fairseq.models as fm

                ref_dc = fm.MODEL_DATACLASS_REGISTRY.get(name) or (fm.MODEL_DATACLASS_REGISTRY.get(owner) if owner else None)
                if ref_dc is None and not dataclasses.is_dataclass(cfg):
                    # nothing to tell a harmless default from an unimplemented option (legacy Namespace configs, aliases without
                    # a reference dataclass): keep the keys this model knows, as before round 3's strict check, and say so
                    import logging

                    known = {f.name for f in dataclasses.fields(cfg_cls)}
                    dropped = sorted(k for k, v in _plain(cfg).items() if k not in known and v is not None)
                    if dropped:
                        logging.getLogger(__name__).warning("model %r: no reference dataclass to check against; ignoring config keys %s",
                                                            name, dropped)
                    cfg = adapt_cfg({k: v for k, v in _plain(cfg).items() if k in known}, cfg_cls, None, "model.")
                else:
                    cfg = adapt_cfg(cfg, cfg_cls, _plain(ref_dc()) if ref_dc is not None else None, "model.")
            elif cfg_cls is None:
                cfg = _plain(cfg)
            return cls.build_model(cfg, inner_task)

    Entry.__name__ = cls.__name__
    Entry.__qualname__ = cls.__qualname__
    return Entry


def criterion_from_cfg(cls, cfg, task):
    """fairseq/criterions/fairseq_criterion.py:28-60 `build_criterion`: constructor arguments by name from the config."""
    d = _plain(cfg)
    kw = {}
    for p in list(inspect.signature(cls.__init__).parameters.values())[1:]:
        if p.name == "task":
            kw["task"] = getattr(task, "inner", task)
        elif p.name in d and d[p.name] is not None:
            kw[p.name] = d[p.name]
        elif p.default is inspect.Parameter.empty and p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY):
            raise ValueError(f"{cls.__name__}: the config has no value for required argument {p.name!r}")
    return cls(**kw)


def install():
    import fairseq.criterions as fc
    import fairseq.models as fm
    import fairseq.tasks as ft
    from fairseq.tasks import FairseqTask

    from .tasks.speech_recognition import SpeechRecognitionEspressoConfig, SpeechRecognitionEspressoTask

    class FairseqTaskAdapter(FairseqTask):
        """FairseqTask facade over `SpeechRecognitionEspressoTask` (same registered name, same call contract)."""

        inner_cls = SpeechRecognitionEspressoTask
        inner_cfg_cls = SpeechRecognitionEspressoConfig

        def __init__(self, cfg, inner):
            super().__init__(cfg)
            self.inner = inner
            self.flat = None

        # -- fairseq_task.py:121 --
        @classmethod
        def setup_task(cls, cfg, **kwargs):
            known = {f.name for f in dataclasses.fields(cls.inner_cfg_cls)}
            d = _plain(cfg)
            ours = cls.inner_cfg_cls(**{k: v for k, v in d.items() if k in known and not (isinstance(v, str) and v.startswith("${"))})
            return cls(cfg, cls.inner_cls.setup_task(ours, **kwargs))

        # -- attributes the reference's models / criterions / CLIs read (espresso/tasks/speech_recognition.py:315-334) --
        feat_dim = property(lambda self: self.inner.feat_dim)
        feat_in_channels = property(lambda self: self.inner.feat_in_channels)
        blank_symbol = property(lambda self: self.inner.blank_symbol)
        word_dictionary = property(lambda self: self.inner.word_dict)
        extra_symbols_to_ignore = property(lambda self: self.inner.extra_symbols_to_ignore)

        @property
        def source_dictionary(self):
            return None

        @property
        def target_dictionary(self):
            return self.inner.target_dictionary

        def max_positions(self):
            return self.inner.max_positions()

        # -- fairseq_task.py:132 / 149 --
        def load_dataset(self, split, epoch=1, combine=False, task_cfg=None, **kwargs):  # espresso/tasks/speech_recognition.py:398
            self.datasets[split] = self.inner.load_dataset(split, epoch=epoch, combine=combine)
            return self.datasets[split]

        # build_model (:327) and build_criterion (:344) are FairseqTask's own: they go through fairseq.models.build_model and
        # fairseq.criterions.build_criterion, i.e. through the registries this module fills.
        def build_model(self, cfg, from_checkpoint=False):
            model = super().build_model(cfg, from_checkpoint)
            if next(model.parameters()).is_cuda or torch.cuda.is_available():
                # the HIP kernels read bf16 shadows of the fp32 masters: keep them in the flat layout and refresh them after
                # every optimizer step of whatever optimizer fairseq builds over these parameters (optimizer_step below)
                from .optim.flat import FlatParams

                self.flat = FlatParams(model.cuda() if not next(model.parameters()).is_cuda else model)
            return model

        # -- fairseq_task.py:359 --
        def build_generator(self, models, args, seq_gen_cls=None, extra_gen_cls_kwargs=None, prefix_allowed_tokens_fn=None):
            return self.inner.build_generator(models, args, seq_gen_cls=seq_gen_cls, extra_gen_cls_kwargs=extra_gen_cls_kwargs)

        # -- fairseq_task.py:490 --
        def _check_ddp(self, model):
            """The native layer runtime accumulates weight gradients straight into p.grad (its autograd backward returns None for
            parameters), so torch's DistributedDataParallel reducer (`--ddp-backend pytorch_ddp`, fairseq's default) never sees
            them as ready: replicas would diverge silently.  fairseq's legacy_ddp / no_c10d wrappers all-reduce the .grad
            buffers after backward (fairseq/distributed/legacy_distributed_data_parallel.py:76-165) and are fine."""
            import torch.distributed as dist

            if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
                return
            from torch.nn.parallel import DistributedDataParallel as TorchDDP

            m, depth = model, 0
            while m is not None and depth < 8:
                if isinstance(m, TorchDDP):
                    raise NotImplementedError(
                        "espresso_amd under fairseq's trainer needs --ddp-backend legacy_ddp (or no_c10d): the pytorch_ddp reducer "
                        "does not see the gradients the HIP layer runtime writes into p.grad")
                m, depth = getattr(m, "module", None), depth + 1

        def train_step(self, sample, model, criterion, optimizer, update_num, ignore_grad=False):
            if not getattr(self, "_ddp_checked", False):
                self._check_ddp(model)
                self._ddp_checked = True
            model.train()
            if hasattr(model, "set_num_updates"):
                model.set_num_updates(update_num)
            sample = self.inner.prepare_sample(sample, train=True)  # fused GPU front-end when the batch carries raw audio
            loss, sample_size, logging_output = criterion(model, sample)
            if ignore_grad:
                loss = loss * 0
            optimizer.backward(loss)
            return loss, sample_size, logging_output

        # -- fairseq_task.py:524 --
        def valid_step(self, sample, model, criterion):
            return self.inner.valid_step(sample, model, criterion)

        # -- fairseq_task.py:530 --
        def optimizer_step(self, optimizer, model, update_num):
            optimizer.step()
            if self.flat is not None:
                self.flat.sync_bf16()

        # -- fairseq_task.py:538 --
        def inference_step(self, generator, models, sample, prefix_tokens=None, constraints=None):
            with torch.no_grad():
                sample = self.inner.prepare_sample(sample, train=False)
                return generator.generate(models, sample, prefix_tokens=prefix_tokens, constraints=constraints)

        def begin_epoch(self, epoch, model):
            self.inner.begin_epoch(epoch, model)

        def reduce_metrics(self, logging_outputs, criterion):
            """fairseq_task.py:564-598 + espresso/tasks/speech_recognition.py:615-629: fairseq's trainer reads the step's statistics
            from its metrics aggregators, so the reduced values (this package's criteria return them as a dict) are logged
            there — words / sentences per batch like FairseqTask does, then every scalar of the reduced dict."""
            out = self.inner.reduce_metrics(logging_outputs, criterion)
            from fairseq.logging import metrics

            if any("ntokens" in log for log in logging_outputs):
                ntokens = sum(log.get("ntokens", 0) for log in logging_outputs)
                metrics.log_scalar("wpb", ntokens, priority=180, round=1)
                metrics.log_speed("wps", ntokens, priority=90, round=1)
            if any("nsentences" in log for log in logging_outputs):
                metrics.log_scalar("bsz", sum(log.get("nsentences", 0) for log in logging_outputs), priority=190, round=1)
            # weights as in the reference's criteria (espresso/criterions/ctc_loss.py:146-160, label_smoothed_cross_entropy_v2.py,
            # transducer_loss.py: loss per sample_size, nll_loss per token) and task (speech_recognition.py:615-629: wer per
            # word, cer per character): fairseq's meters average log_scalar values by weight over an epoch / validation pass,
            # and --best-checkpoint-metric / reduce_lr_on_plateau read those averages
            tot = lambda k: sum(log.get(k, 0) for log in logging_outputs)
            weights = {"loss": tot("sample_size"), "nll_loss": tot("ntokens"), "wer": tot("word_count"), "cer": tot("char_count")}
            for k, v in (out or {}).items():
                if not (isinstance(v, (int, float)) or (torch.is_tensor(v) and v.numel() == 1)):
                    continue
                if k in ("sample_size", "ppl", "word_error", "word_count", "char_error", "char_count"):
                    continue  # (ppl is derived below; the reference logs neither the raw error counts nor sample_size)
                if k in ("wer", "cer") and not weights[k]:
                    continue  # (no reference words / characters in these batches: the reference logs no error rate then)
                if k in weights:
                    metrics.log_scalar(k, float(v), float(weights[k]) if weights[k] else 1, round=4 if k in ("wer", "cer") else 3)
                else:
                    metrics.log_scalar(k, float(v), round=3)
            if out and ("nll_loss" in out or "ppl" in out):
                from fairseq import utils as fq_utils

                key = "nll_loss" if "nll_loss" in out else "loss"
                metrics.log_derived("ppl", lambda meters, key=key: fq_utils.get_perplexity(meters[key].avg))
            return out

    FairseqTaskAdapter.__name__ = "SpeechRecognitionEspressoTask"
    name = "speech_recognition_espresso"
    ft.TASK_REGISTRY[name] = FairseqTaskAdapter  # (TASK_DATACLASS_REGISTRY keeps the reference's dataclass: fairseq merges into it)

    for mname, cls in registry.MODEL_REGISTRY.items():
        entry = _model_entry(mname, cls)
        fm.MODEL_REGISTRY[mname] = entry
        fm.ARCH_MODEL_REGISTRY[mname] = entry  # fairseq/models/__init__.py:78: checked first
        fm.ARCH_MODEL_NAME_REGISTRY[mname] = mname
    for aname, cls in registry.ARCH_MODEL_REGISTRY.items():
        owner = next((n for n, c in registry.MODEL_REGISTRY.items() if c is cls), aname)
        fm.ARCH_MODEL_REGISTRY[aname] = fm.MODEL_REGISTRY.get(owner, _model_entry(aname, cls, owner=owner))
        fm.ARCH_MODEL_NAME_REGISTRY[aname] = owner
        fm.ARCH_CONFIG_REGISTRY[aname] = registry.ARCH_CONFIG_REGISTRY[aname]

    for cname, cls in registry.CRITERION_REGISTRY.items():
        if not hasattr(cls, "build_criterion"):
            cls.build_criterion = classmethod(criterion_from_cfg)
        fc.CRITERION_REGISTRY[cname] = cls
    return FairseqTaskAdapter
