"""Batched recognition CLI — the decoding loop and output format of espresso/speech_recognize.py:60-360 on the HIP path:
for every batch run the chosen search (beam search with optional LM / look-ahead word-LM fusion, CTC greedy, transducer
greedy / beam), print `T-<utt>` (reference) and `H-<utt>` (hypothesis, score in base 2) lines, accumulate WER / CER with
`tools.wer.Scorer`, and close with the "Recognized N utterances ..." summary.

Checkpoint management is fairseq's and stays out of this framework (SURVEY §2 out of scope): the model is built from a
config mapping (`--model-config`, the `model:` block of the recipe YAML as JSON/YAML) and a `state_dict` file (`--path`:
either a plain state_dict or a fairseq checkpoint dict whose `"model"` entry is the state_dict — reference checkpoints load
because the parameter names are identical).  Audio comes from a Kaldi-style `wav.scp` (`utt_id path.wav`, 16-bit PCM read
with the stdlib) and optional `text` (`utt_id tokens...`) files; the front-end (fbank + CMVN) runs on the GPU."""
import argparse
import os
import json
import math
import sys
import time
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch


from .data.audio_utils import read_wav  # noqa: E402,F401


def read_scp(path: str) -> Dict[str, str]:
    out = {}
    for line in open(path, encoding="utf-8"):
        line = line.strip()
        if line:
            k, v = line.split(None, 1)
            out[k] = v
    return out


def make_batches(utt_ids: List[str], n_samples: List[int], max_tokens: int, max_sentences: int) -> List[List[int]]:
    """Length-sorted batches under the frame budget (frames = samples / 160), longest first inside a batch."""
    order = sorted(range(len(utt_ids)), key=lambda i: -n_samples[i])
    batches, cur, mx = [], [], 0
    for i in order:
        fr = n_samples[i] // 160 + 1
        if cur and (max(mx, fr) * (len(cur) + 1) > max_tokens or len(cur) + 1 > max_sentences):
            batches.append(cur)
            cur, mx = [], 0
        cur.append(i)
        mx = max(mx, fr)
    if cur:
        batches.append(cur)
    return batches


def shard_batches(batches: List[List[int]], num_shards: int, shard_id: int) -> List[List[int]]:
    """The batches of one decoding replica: espresso/speech_recognize.py:188-189 hands `num_shards = distributed_world_size`,
    `shard_id = distributed_rank` to the batch iterator, whose ShardedIterator (fairseq/data/iterators.py:534-571) deals the
    length-sorted batch list round-robin: replica i decodes batches i, i + n, i + 2n, ... (no collective: SURVEY 8(e) "replicas
    only"); every replica scores its own shard."""
    if not (0 <= shard_id < num_shards):
        raise ValueError(f"--shard-id {shard_id} outside [0, --num-shards {num_shards})")
    return batches[shard_id::num_shards]


def recognize(task, model, generator, batches: Iterable[dict], dictionary, refs: Optional[Dict[str, str]] = None, out=sys.stdout,
              nbest: int = 1, quiet: bool = False, bpe_symbol=None):
    """The loop of espresso/speech_recognize.py:226-330.  `batches` yield dicts with `utt_ids`, `wav`, `wav_offsets`,
    `num_samples` (device tensors / lists as produced by `collate`).  Returns (scorer, stats)."""
    from .tools.wer import Scorer

    scorer = Scorer(dictionary, wer_output_filter=None)
    num_sent, num_tok, t_gen, audio_s = 0, 0, 0.0, 0.0
    for sample in batches:
        t0 = time.perf_counter()
        s = task.prepare_sample(sample, train=False)
        hypos = generator.generate([model], s)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        t_gen += time.perf_counter() - t0
        audio_s += sum(sample["num_samples"]) / 16000.0
        for i, utt in enumerate(sample["utt_ids"]):
            if refs is not None and utt in refs and not quiet:
                print("T-{}\t{}".format(utt, refs[utt]), file=out)
            for j, hypo in enumerate(hypos[i][:nbest]):
                toks = hypo["tokens"].int().cpu()
                strip = getattr(generator, "symbols_to_strip_from_output", None) or {dictionary.eos(), dictionary.pad()}
                hypo_str = dictionary.string(torch.tensor([t for t in toks.tolist() if t not in strip]), bpe_symbol=bpe_symbol)
                if not quiet:
                    print("H-{}\t{}\t{}".format(utt, hypo_str, float(hypo["score"]) / math.log(2)), file=out)
                if j == 0:
                    scorer.add_prediction(utt, hypo_str)
                    if refs is not None and utt in refs:
                        scorer.add_evaluation(utt, refs[utt], hypo_str)
                    num_tok += len(toks)
        num_sent += len(sample["utt_ids"])
    print("NOTE: hypothesis and token scores are output in base 2", file=out)
    print("Recognized {:,} utterances ({} tokens) in {:.1f}s ({:.2f} sentences/s, {:.2f} tokens/s), RTF {:.4f}".format(
        num_sent, num_tok, t_gen, num_sent / max(t_gen, 1e-9), num_tok / max(t_gen, 1e-9), t_gen / max(audio_s, 1e-9)), file=out)
    if refs:
        for line in scorer.summary_lines():
            print(line, file=out)
    return scorer, {"sentences": num_sent, "tokens": num_tok, "seconds": t_gen, "rtf": t_gen / max(audio_s, 1e-9)}


def collate(ids: List[int], utt_ids: List[str], waves: List[np.ndarray], device):
    lens = [len(waves[i]) for i in ids]
    offs = np.zeros(len(ids) + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    return {"utt_ids": [utt_ids[i] for i in ids], "wav": torch.from_numpy(np.concatenate([waves[i] for i in ids])).to(device),
            "wav_offsets": torch.from_numpy(offs).to(device), "num_samples": lens, "net_input": {}}


def build_generator(args, model, dictionary, lm=None):
    from .sequence_generator import SequenceGenerator
    from .tools.ctc_decoder import CTCDecoder
    from .tools.transducer_beam_search_decoder import TransducerBeamSearchDecoder
    from .tools.transducer_greedy_decoder import TransducerGreedyDecoder

    if args.search == "ctc":
        return CTCDecoder([model], dictionary)
    if args.search == "transducer_greedy":
        return TransducerGreedyDecoder([model], dictionary, max_num_expansions_per_step=args.max_num_expansions_per_step,
                                       lm_model=lm, lm_weight=args.lm_weight)
    if args.search == "transducer_beam":
        return TransducerBeamSearchDecoder([model], dictionary, beam_size=args.beam,
                                           max_num_expansions_per_step=args.max_num_expansions_per_step, expansion_beta=args.expansion_beta,
                                           expansion_gamma=args.expansion_gamma, prefix_alpha=args.prefix_alpha, lm_model=lm,
                                           lm_weight=args.lm_weight)
    return SequenceGenerator(model if isinstance(model, (list, tuple)) else [model], dictionary, beam_size=args.beam, max_len_a=args.max_len_a, max_len_b=args.max_len_b,
                             min_len=args.min_len, normalize_scores=not args.unnormalized, len_penalty=args.lenpen,
                             unk_penalty=args.unkpen, temperature=args.temperature, lm_model=lm, lm_weight=args.lm_weight,
                             eos_factor=args.eos_factor)


def get_parser():
    p = argparse.ArgumentParser("espresso_amd.speech_recognize", description=__doc__.split("\n")[0])
    p.add_argument("--path", required=True, help="state_dict (or fairseq checkpoint dict with a 'model' entry)")
    p.add_argument("--model", default=None, help="registered model name (default: the checkpoint's cfg.model._name, else speech_transformer_base)")
    p.add_argument("--model-config", default=None,
                   help="JSON/YAML file with the recipe's `model:` block (default: the `cfg.model` stored in the checkpoint, as the "
                        "reference rebuilds the model in checkpoint_utils.load_model_ensemble)")
    p.add_argument("--dict", required=True)
    p.add_argument("--wav-scp", required=True)
    p.add_argument("--text", default=None, help="reference transcripts (utt_id tokens...)")
    p.add_argument("--global-cmvn-stats-path", default=None)
    p.add_argument("--search", default="beam", choices=["beam", "ctc", "transducer_greedy", "transducer_beam"])
    p.add_argument("--beam", type=int, default=10)
    p.add_argument("--nbest", type=int, default=1)
    p.add_argument("--max-len-a", type=float, default=0.08)
    p.add_argument("--max-len-b", type=int, default=0)
    p.add_argument("--min-len", type=int, default=1)
    p.add_argument("--unnormalized", action="store_true")
    p.add_argument("--lenpen", type=float, default=1.0)
    p.add_argument("--unkpen", type=float, default=0.0)
    p.add_argument("--temperature", type=float, default=1.0)
    p.add_argument("--eos-factor", type=float, default=None)
    p.add_argument("--lm-path", default=None)
    p.add_argument("--lm-arch", default="lstm_lm_librispeech")
    p.add_argument("--lm-weight", type=float, default=0.0)
    p.add_argument("--word-dict", default=None, help="enables look-ahead word-LM fusion (the LM at --lm-path is a word LM)")
    p.add_argument("--oov-penalty", type=float, default=1e-4)
    p.add_argument("--max-num-expansions-per-step", type=int, default=2)
    p.add_argument("--expansion-beta", type=int, default=0)
    p.add_argument("--expansion-gamma", type=float, default=None)
    p.add_argument("--prefix-alpha", type=int, default=None)
    p.add_argument("--max-tokens", type=int, default=15000)
    p.add_argument("--batch-size", type=int, default=24)
    p.add_argument("--quiet", action="store_true")
    p.add_argument("--num-shards", type=int, default=int(os.environ.get("WORLD_SIZE", "1")),
                   help="decode replicas (default: WORLD_SIZE); each takes every num-shards-th batch")
    p.add_argument("--shard-id", type=int, default=int(os.environ.get("RANK", "0")), help="this replica (default: RANK)")
    p.add_argument("--device", default=None, help="default: cuda:LOCAL_RANK (cuda:0 outside a launcher)")
    return p


def _load_file(path):
    try:
        return torch.load(path, map_location="cpu")
    except Exception:  # checkpoints written by the reference carry Namespace / omegaconf objects next to the tensors
        return torch.load(path, map_location="cpu", weights_only=False)


def _load_state(path):
    sd = _load_file(path)
    return sd["model"] if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict) else sd


def resolve_model_config(model_name, model_config_path, checkpoint):
    """(registered model name, `model:` block as a dict): explicit arguments win; otherwise what the checkpoint's `cfg` holds
    (fairseq/checkpoint_utils.py:422-470 rebuilds the model from `state["cfg"].model` the same way)."""
    import yaml

    stored = None
    cfg = checkpoint.get("cfg") if isinstance(checkpoint, dict) else None
    if cfg is not None:
        stored = cfg["model"] if isinstance(cfg, dict) else getattr(cfg, "model", None)
        if stored is not None and not isinstance(stored, dict):
            try:
                from omegaconf import OmegaConf  # a reference checkpoint opened where omegaconf exists

                stored = OmegaConf.to_container(stored, resolve=True)
            except ImportError:
                stored = dict(vars(stored)) if hasattr(stored, "__dict__") else dict(stored)
    if model_config_path:
        block = json.load(open(model_config_path)) if model_config_path.endswith(".json") else yaml.safe_load(open(model_config_path))
        block = block.get("model", block) if isinstance(block.get("model"), dict) else block  # a whole recipe file is fine too
    elif stored is not None:
        block = dict(stored)
    else:
        raise ValueError("--model-config is required: the checkpoint holds no cfg.model block")
    name = model_name or block.get("_name") or (stored or {}).get("_name") or "speech_transformer_base"
    return name, block


def main(argv=None):
    args = get_parser().parse_args(argv)
    import yaml

    from . import registry
    from .data.asr_dictionary import AsrDictionary
    from .models.lstm_lm import LSTMLanguageModelEspresso
    from .models.tensorized_lookahead_language_model import TensorizedLookaheadLanguageModel
    from .tasks.speech_recognition import SpeechRecognitionEspressoConfig, SpeechRecognitionEspressoTask

    dev = torch.device(args.device or "cuda:{}".format(int(os.environ.get("LOCAL_RANK", "0"))))
    if dev.type == "cuda":
        if dev.index is None:  # `--device cuda`: set_device needs an explicit index
            dev = torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(dev)
    paths = args.path.split(os.pathsep)  # `--path a.pt:b.pt` = an ensemble (fairseq utils.split_paths in speech_recognize.py:107)
    state = _load_file(paths[0])
    model_name, model_cfg = resolve_model_config(args.model, args.model_config, state)
    autoregressive = args.search == "beam"
    # the criterion the checkpoint was trained with decides whether "<s>" is the blank (speech_recognition.py:324, 345-347)
    crit = {"beam": "label_smoothed_cross_entropy_v2", "ctc": "ctc_loss"}.get(args.search, "transducer_loss")
    task = SpeechRecognitionEspressoTask.setup_task(SpeechRecognitionEspressoConfig(
        dict=args.dict, autoregressive=autoregressive, global_cmvn_stats_path=args.global_cmvn_stats_path, criterion_name=crit))
    def load_member(state, name, block):
        cls = registry.MODEL_REGISTRY[name]
        cfg_cls = getattr(cls, "config_class", None)
        cfg = cfg_cls.from_dict(block) if cfg_cls is not None else block
        m = cls.build_model(cfg, task)
        sd = state["model"] if isinstance(state, dict) and isinstance(state.get("model"), dict) else state
        if hasattr(m, "upgrade_state_dict_named"):
            sd = m.upgrade_state_dict_named(dict(sd), "")
        m.load_state_dict(sd, strict=True)
        return m.to(dev).eval()

    model = load_member(state, model_name, model_cfg)
    members = [model]
    for extra in paths[1:]:  # every member is rebuilt from ITS OWN checkpoint's configuration (checkpoint_utils.load_model_ensemble)
        st = _load_file(extra)
        members.append(load_member(st, *resolve_model_config(args.model, args.model_config, st)))
    if len(members) > 1 and args.search != "beam":
        raise NotImplementedError("ensembles are implemented for the attention decoder's beam search (--search beam)")
    lm = None
    if args.lm_path:
        class _LMTask:
            target_dictionary = source_dictionary = task.target_dictionary
        if args.word_dict:
            _LMTask.word_dictionary = AsrDictionary.load(args.word_dict, enable_bos=False)
        lm = LSTMLanguageModelEspresso.build_model(dict(arch=args.lm_arch, is_wordlm=bool(args.word_dict)), _LMTask)
        lm.load_state_dict(_load_state(args.lm_path), strict=True)
        lm = lm.to(dev).eval()
        if args.word_dict:
            lm = TensorizedLookaheadLanguageModel(lm, task.target_dictionary, oov_penalty=args.oov_penalty)
    gen = build_generator(args, members if len(members) > 1 else model, task.target_dictionary, lm)
    scp = read_scp(args.wav_scp)
    utt_ids = list(scp.keys())
    waves = [read_wav(scp[u]) for u in utt_ids]
    refs = read_scp(args.text) if args.text else None
    task.build_frontend(dev)
    batches = shard_batches(make_batches(utt_ids, [len(w) for w in waves], args.max_tokens, args.batch_size), args.num_shards, args.shard_id)
    if args.num_shards > 1:  # (the defaults come from WORLD_SIZE / RANK: say that this process decodes — and scores — a part only)
        print(f"| decoding shard {args.shard_id} of {args.num_shards}: {sum(len(b) for b in batches)} of {len(utt_ids)} utterances; "
              "WER / CER below cover this shard only", file=sys.stderr)
    recognize(task, model, gen, (collate(b, utt_ids, waves, dev) for b in batches), task.target_dictionary, refs, out=sys.stdout,
              nbest=args.nbest, quiet=args.quiet)


if __name__ == "__main__":
    main()
