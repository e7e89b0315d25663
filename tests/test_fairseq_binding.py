"""Drop-in boundary from the REFERENCE's side: with fairseq + espresso importable (here: /root/reference through
the stub packages of oracle/ref_stubs — the build container only; skipped on the GPU box, which has neither),
`registry.mirror_into_fairseq()` makes fairseq's own entry points resolve to espresso_amd classes with the reference's call
contracts: fairseq/tasks/__init__.py:26 setup_task, fairseq_task.py:327 build_model -> fairseq/models/__init__.py:55,
fairseq_task.py:344 build_criterion -> fairseq/criterions (registry build_x), and the FairseqTask method signatures the
trainer / speech_recognize.py call (fairseq_task.py:132,359,490,524,530,538)."""
import inspect
import os
import sys

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (build container only)")


@pytest.fixture(scope="module")
def fairseq_env(tmp_path_factory):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "oracle", "ref_stubs"))
    sys.path.insert(1, REF)
    import fairseq  # noqa: F401
    import espresso  # noqa: F401  (registers the REFERENCE's classes under the names below)
    from fairseq import criterions, models, tasks

    ref_task_cls = tasks.TASK_REGISTRY["speech_recognition_espresso"]
    ref_model_cls = models.MODEL_REGISTRY["speech_transformer_encoder_model"]
    ref_crit_cls = criterions.CRITERION_REGISTRY["ctc_loss"]
    import espresso_amd  # noqa: F401
    from espresso_amd import registry

    adapter = registry.mirror_into_fairseq()
    tmp = tmp_path_factory.mktemp("data")
    with open(tmp / "dict.txt", "w") as f:
        f.write("".join(f"t{i} 1\n" for i in range(30)))
    return dict(tasks=tasks, models=models, criterions=criterions, adapter=adapter, tmp=str(tmp), ref=(ref_task_cls, ref_model_cls, ref_crit_cls))


def _cfg(d):
    from omegaconf import DictConfig

    return DictConfig({k: (_cfg(v) if isinstance(v, dict) else v) for k, v in d.items()})


def test_setup_task_build_model_build_criterion_resolve_to_espresso_amd(fairseq_env):
    e = fairseq_env
    import torch

    from espresso_amd.criterions.ctc_loss import CtcLossCriterion
    from espresso_amd.models.transformer.speech_transformer_encoder_model import SpeechTransformerEncoderModel
    from espresso_amd.tasks.speech_recognition import SpeechRecognitionEspressoTask
    from fairseq.tasks import FairseqTask

    # the registries no longer hand out the reference's classes
    assert e["tasks"].TASK_REGISTRY["speech_recognition_espresso"] is e["adapter"] is not e["ref"][0]
    assert issubclass(e["models"].ARCH_MODEL_REGISTRY["speech_transformer_encoder_model"], SpeechTransformerEncoderModel)
    assert e["criterions"].CRITERION_REGISTRY["ctc_loss"] is CtcLossCriterion is not e["ref"][2]

    # fairseq.tasks.setup_task: the user's config is merged into the REFERENCE's dataclass, then our task is set up from it
    task = e["tasks"].setup_task(_cfg({"_name": "speech_recognition_espresso", "data": e["tmp"], "dict": os.path.join(e["tmp"], "dict.txt"),
                                       "criterion_name": "ctc_loss", "max_source_positions": 3000}))
    assert isinstance(task, FairseqTask) and isinstance(task.inner, SpeechRecognitionEspressoTask)
    assert task.blank_symbol == "<s>" and len(task.target_dictionary) == 30 + 4 and task.max_positions()[0] == 3000
    assert task.feat_dim == 80 and task.source_dictionary is None

    # task.build_model(cfg.model) -> fairseq.models.build_model -> our class, built from the reference-style nested config
    model = task.build_model(_cfg({"_name": "speech_transformer_encoder_model",
                                   "encoder": {"embed_dim": 128, "ffn_embed_dim": 256, "layers": 2, "attention_heads": 2, "layer_type": "conformer",
                                               "normalize_before": True, "relative_positional_embeddings": True,
                                               "conv_channels": "[64, 64, 16, 16]"},
                                   "layernorm_embedding": True, "dropout": 0.1}))
    assert isinstance(model, SpeechTransformerEncoderModel) and isinstance(model, torch.nn.Module)
    assert model.cfg.encoder.embed_dim == 128 and model.cfg.encoder.layer_type == "conformer" and model.cfg.dropout == 0.1
    keys = set(model.state_dict())
    assert {"encoder.fc0.weight", "encoder.layers.1.self_attn.pos_bias_u", "encoder.layers.0.conv_module.depthwise_conv.weight",
            "encoder.fc_out.weight"} <= keys  # the reference's parameter names (checkpoints move both ways)

    # a configuration value the HIP path does not implement is refused, not dropped
    with pytest.raises(NotImplementedError):
        task.build_model(_cfg({"_name": "speech_transformer_encoder_model", "quant_noise": {"pq": 0.1}}))

    # task.build_criterion(cfg.criterion) -> fairseq.criterions.build_criterion -> our class with the config's values
    crit = task.build_criterion(_cfg({"_name": "ctc_loss", "sentence_avg": False, "zero_infinity": True, "print_training_sample_interval": 77}))
    assert type(crit) is CtcLossCriterion and crit.sentence_avg is False and crit.print_interval == 77
    assert crit.blank_idx == task.target_dictionary.index("<s>")


def test_task_adapter_keeps_the_fairseq_task_signatures(fairseq_env):
    """Positional parameter names of every method fairseq's trainer / CLIs call on a task."""
    from fairseq.tasks import FairseqTask

    A = fairseq_env["adapter"]
    ref_task = fairseq_env["ref"][0]
    # load_dataset: the espresso task's own signature (speech_recognition.py:398: split, epoch, combine) plus FairseqTask's task_cfg
    assert list(inspect.signature(ref_task.load_dataset).parameters)[:4] == list(inspect.signature(A.load_dataset).parameters)[:4]
    for name in ("build_model", "build_criterion", "build_generator", "train_step", "valid_step", "optimizer_step",
                 "inference_step", "begin_epoch", "reduce_metrics", "max_positions", "get_batch_iterator", "dataset",
                 "filter_indices_by_size", "state_dict", "load_state_dict"):
        want = [p for p in inspect.signature(getattr(FairseqTask, name)).parameters]
        got = [p for p in inspect.signature(getattr(A, name)).parameters]
        assert got[: len(want)] == want or set(want) <= set(got), (name, want, got)
    want = list(inspect.signature(FairseqTask.train_step).parameters)
    assert list(inspect.signature(A.train_step).parameters) == want == ["self", "sample", "model", "criterion", "optimizer", "update_num", "ignore_grad"]


def test_other_registered_names_resolve(fairseq_env):
    e = fairseq_env
    from espresso_amd import registry

    for name, cls in registry.MODEL_REGISTRY.items():
        assert issubclass(e["models"].MODEL_REGISTRY[name], cls), name
    for name, cls in registry.CRITERION_REGISTRY.items():
        assert e["criterions"].CRITERION_REGISTRY[name] is cls and hasattr(cls, "build_criterion"), name
    for arch in registry.ARCH_MODEL_REGISTRY:
        assert arch in e["models"].ARCH_MODEL_REGISTRY and arch in e["models"].ARCH_CONFIG_REGISTRY, arch


def test_adapt_cfg_refuses_what_it_would_drop(fairseq_env):
    """a non-default value of a field the HIP path does not implement raises — including a key the reference's own
    dataclass does not have (nothing says it is safe to ignore); None / `${...}` interpolations / reference defaults pass"""
    import dataclasses

    from espresso_amd.fairseq_plugin import adapt_cfg

    @dataclasses.dataclass
    class Ours:
        a: int = 1

    ref = {"a": 1, "b": 2.0, "c": None}
    assert adapt_cfg({"a": 5, "b": 2.0, "c": None, "d": None, "e": "${task.x}"}, Ours, ref).a == 5
    with pytest.raises(NotImplementedError):
        adapt_cfg({"a": 5, "b": 3.0}, Ours, ref)          # non-default value of an unimplemented reference field
    with pytest.raises(NotImplementedError):
        adapt_cfg({"a": 5, "zzz": 1}, Ours, ref)          # a key nobody knows


def test_task_adapter_refuses_pytorch_ddp(fairseq_env, monkeypatch):
    """the native layer runtime writes weight gradients into p.grad behind autograd's back; torch's DDP reducer never
    sees them, so the adapter refuses a model wrapped in it when world_size > 1 (legacy_ddp / no_c10d all-reduce .grad buffers)"""
    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as TorchDDP

    adapter = fairseq_env["adapter"]
    task = adapter.__new__(adapter)
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda *a, **k: 2)
    wrapped = TorchDDP.__new__(TorchDDP)   # no process group needed: only the type is inspected

    class Proxy:  # fairseq's ModuleProxyWrapper keeps the wrapped module in `.module`
        def __init__(self, m):
            self.module = m

    with pytest.raises(NotImplementedError, match="legacy_ddp"):
        task._check_ddp(Proxy(wrapped))
    task._check_ddp(Proxy(torch.nn.Linear(2, 2)))  # anything else passes


def test_train_step_through_fairseqs_own_trainer(fairseq_env):
    """fairseq/trainer.py:801-953 `Trainer.train_step` driving the adapter task for five updates: fairseq's optimizer (FairseqAdam
    built by fairseq over the model's parameters), its gradient clipping, its logging aggregation — with a model whose weight
    gradient is written into `p.grad` behind autograd's back and never returned to autograd, which is what the HIP layer runtime
    does (the real layers need the GPU; this stand-in keeps the contract and runs on the CPU).  Checks: the loss falls, the
    behind-the-back parameter is updated by fairseq's optimizer, the update counter advances, the task's reduce_metrics feeds
    fairseq's aggregators (wpb / bsz), and the gradient norm fairseq clips with includes that parameter."""
    e = fairseq_env
    import torch
    from fairseq.criterions import FairseqCriterion
    from fairseq.dataclass.configs import FairseqConfig
    from fairseq.logging import metrics
    from fairseq.models import BaseFairseqModel
    from fairseq.trainer import Trainer

    task = e["tasks"].setup_task(_cfg({"_name": "speech_recognition_espresso", "data": e["tmp"], "dict": os.path.join(e["tmp"], "dict.txt"),
                                       "criterion_name": "ctc_loss"}))

    class BehindAutograd(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, holder):
            ctx.save_for_backward(x)
            ctx.holder = holder
            return x @ holder.W.detach().t()

        @staticmethod
        def backward(ctx, dy):
            (x,) = ctx.saved_tensors
            W = ctx.holder.W
            if W.grad is None:
                W.grad = torch.zeros_like(W)
            W.grad += dy.t() @ x  # accumulated in place, nothing returned for W: the native runtime's pattern
            return dy @ W.detach(), None

    class Model(BaseFairseqModel):
        def __init__(self):
            super().__init__()
            self.inp = torch.nn.Linear(8, 8)
            self.W = torch.nn.Parameter(torch.randn(4, 8) * 0.3)

        def forward(self, src_tokens, **kw):
            return BehindAutograd.apply(torch.tanh(self.inp(src_tokens)), self)

    class Crit(FairseqCriterion):
        def forward(self, model, sample, reduce=True):
            loss = ((model(**sample["net_input"]) - sample["target"]) ** 2).sum()
            n = sample["target"].shape[0]
            return loss, n, {"loss": loss.detach(), "ntokens": n, "nsentences": n, "sample_size": n}

        @staticmethod
        def reduce_metrics(logging_outputs):
            metrics.log_scalar("loss", sum(float(log["loss"]) for log in logging_outputs))

        @staticmethod
        def logging_outputs_can_be_summed():
            return True

    torch.manual_seed(0)
    cfg = FairseqConfig()
    cfg.common.cpu = True
    cfg.optimizer = _cfg({"_name": "adam", "lr": [0.01]})
    cfg.lr_scheduler = _cfg({"_name": "fixed", "lr": [0.01]})
    cfg.optimization.lr = [0.01]
    cfg.optimization.clip_norm = 1.0
    model, crit = Model(), Crit(task)
    trainer = Trainer(cfg, task, model, crit)
    X, T = torch.randn(16, 8), torch.randn(16, 4)
    sample = {"net_input": {"src_tokens": X}, "target": T, "ntokens": 16, "id": torch.arange(16)}
    w0 = model.W.detach().clone()
    losses, gnorms = [], []
    with metrics.aggregate("train_inner") as agg:
        for _ in range(5):
            trainer.train_step([sample])
            with torch.no_grad():
                losses.append(float(((model(X) - T) ** 2).sum()))
        logged = agg.get_smoothed_values()
    assert all(b < a for a, b in zip(losses, losses[1:])), losses
    assert float((model.W - w0).abs().max()) > 1e-3          # fairseq's optimizer stepped the behind-the-back parameter
    assert trainer.get_num_updates() == 5
    assert logged.get("wpb") == 16 and logged.get("bsz") == 16 and "loss" in logged, logged
    # the norm fairseq clips with covers W: the same step's gradient by hand (fairseq scales by 1 / sample_size before clipping)
    trainer.zero_grad()
    crit(model, sample)[0].backward()
    assert model.W.grad is not None and float(model.W.grad.norm()) > 0
    with_w = float(torch.sqrt(sum((p.grad ** 2).sum() for p in model.parameters() if p.grad is not None))) / 16
    without_w = float(torch.sqrt(sum((p.grad ** 2).sum() for n, p in model.named_parameters() if p.grad is not None and n != "W"))) / 16
    trainer.zero_grad()
    with metrics.aggregate("one_step") as agg:
        trainer.train_step([sample])
        gnorm = float(agg.get_smoothed_values()["gnorm"])
    assert abs(gnorm - with_w) <= 2e-3 * with_w + 1e-3, (gnorm, with_w, without_w)
    assert abs(gnorm - without_w) > 10 * (2e-3 * with_w + 1e-3), (gnorm, with_w, without_w)
    # fairseq/trainer.py:1086-1150 valid_step -> task.valid_step(sample, model, criterion): eval mode, no gradient, statistics back
    w1 = model.W.detach().clone()
    stats = trainer.valid_step(sample)
    assert "loss" in stats and float((model.W - w1).abs().max()) == 0.0 and not model.training


def test_adapter_reduce_metrics_weights_match_the_reference(fairseq_env):
    """over an epoch / validation pass fairseq's meters average `log_scalar` values by WEIGHT.  The reference logs loss
    per sample_size, nll_loss per token (espresso/criterions/ctc_loss.py:146-160), wer per word and cer per character
    (espresso/tasks/speech_recognition.py:615-629) — so the aggregated values are corpus-level ratios, not means of per-batch
    ratios (they drive --best-checkpoint-metric and reduce_lr_on_plateau).  Two batches of very different size through the
    adapter's reduce_metrics with this package's ctc_loss criterion."""
    import math

    e = fairseq_env
    from fairseq.logging import metrics

    from espresso_amd.criterions.ctc_loss import CtcLossCriterion

    task = e["tasks"].setup_task(_cfg({"_name": "speech_recognition_espresso", "data": e["tmp"], "dict": os.path.join(e["tmp"], "dict.txt"),
                                       "criterion_name": "ctc_loss"}))
    logs = [{"loss": 100.0, "ntokens": 50, "nsentences": 2, "sample_size": 2, "word_error": 1, "word_count": 10, "char_error": 2, "char_count": 40},
            {"loss": 9000.0, "ntokens": 950, "nsentences": 30, "sample_size": 30, "word_error": 90, "word_count": 190, "char_error": 300, "char_count": 760}]
    with metrics.aggregate("valid_pass") as agg:
        for log in logs:
            task.reduce_metrics([log], CtcLossCriterion)
        v = agg.get_smoothed_values()
    assert v["loss"] == pytest.approx(9100.0 / 32 / math.log(2), rel=1e-3)       # not the mean of 50/ln2 and 300/ln2
    assert v["nll_loss"] == pytest.approx(9100.0 / 1000 / math.log(2), rel=1e-3)
    assert v["wer"] == pytest.approx(91 / 200 * 100, rel=1e-3) and v["cer"] == pytest.approx(302 / 800 * 100, rel=1e-3)
    assert v["ppl"] == pytest.approx(2 ** (9100.0 / 1000 / math.log(2)), rel=1e-2)
    assert "sample_size" not in v and "word_count" not in v


# ---- fairseq's legacy_ddp wrapper over gradients written behind autograd's back, two ranks (gloo) ---------------------------------
def test_fairseq_legacy_ddp_reduces_gradients_written_behind_autograd(fairseq_env):
    """`--ddp-backend legacy_ddp` (fairseq/distributed/legacy_distributed_data_parallel.py:76-165), which INTEGRATION.md prescribes
    under fairseq's trainer: its explicit all_reduce_grads() reads `p.grad` after backward, so a weight gradient the layer runtime
    accumulated in place — never returned to autograd — is averaged over the ranks like every other (world 2, gloo, with one
    no_sync micro-batch; torch's reducer-based DDP is the backend the adapter refuses).  Ranks run tests/fairseq_legacy_ddp_worker.py."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fairseq_legacy_ddp_worker.py")
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", str(port)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
