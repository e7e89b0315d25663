"""Beam-search semantics pinned by the reference's own known-answer test (tests/test_sequence_generator.py:166-330 with the
scripted decoder of tests/utils.py:67-165): exact hypotheses and scores for beam 2 with / without score normalisation and
with length penalties.  CPU: host logic + oracle search.  GPU (-m gpu): the same answers with the HIP beam kernels."""
import pytest
import torch

from espresso_amd.sequence_generator import HipBeamSearch, SequenceGenerator
from oracle.search_ref import TorchRefSearch, scripted_setup


def _check(h, tokens, pos_probs, normalized=True, lenpen=1.0):
    assert h["tokens"].tolist() == tokens
    ps = torch.FloatTensor(pos_probs).log()
    assert float((h["positional_scores"].float().cpu() - ps).abs().max()) < 1e-4
    score = ps.sum()
    if normalized:
        score = score / ps.numel() ** lenpen
    assert abs(float(score) - float(h["score"])) < 1e-5


def _run(search, device, **kw):
    d, w1, w2, sample, model = scripted_setup()
    sample = {"net_input": {k: v.to(device) for k, v in sample["net_input"].items()}}
    gen = SequenceGenerator([model], d, beam_size=2, search=search, **kw)
    return d.eos(), w1, w2, gen.generate([model], sample)


def _assert_scenarios(search, device):
    eos, w1, w2, hypos = _run(search, device)
    _check(hypos[0][0], [w1, eos], [0.9, 1.0])
    _check(hypos[0][1], [w2, w1, w2, eos], [0.1, 0.9, 0.9, 1.0])
    _check(hypos[1][0], [w1, w2, w1, eos], [0.7, 0.4, 0.4, 1.0])
    _check(hypos[1][1], [w1, w2, eos], [0.7, 0.4, 0.6])
    eos, w1, w2, hypos = _run(search, device, normalize_scores=False)
    _check(hypos[0][0], [w1, eos], [0.9, 1.0], normalized=False)
    _check(hypos[0][1], [w2, w1, w2, eos], [0.1, 0.9, 0.9, 1.0], normalized=False)
    _check(hypos[1][0], [w1, w2, eos], [0.7, 0.4, 0.6], normalized=False)
    _check(hypos[1][1], [w1, w2, w1, eos], [0.7, 0.4, 0.4, 1.0], normalized=False)
    eos, w1, w2, hypos = _run(search, device, len_penalty=0.6)
    _check(hypos[0][0], [w1, eos], [0.9, 1.0], lenpen=0.6)
    _check(hypos[0][1], [w2, w1, w2, eos], [0.1, 0.9, 0.9, 1.0], lenpen=0.6)
    _check(hypos[1][0], [w1, w2, eos], [0.7, 0.4, 0.6], lenpen=0.6)
    _check(hypos[1][1], [w1, w2, w1, eos], [0.7, 0.4, 0.4, 1.0], lenpen=0.6)
    eos, w1, w2, hypos = _run(search, device, len_penalty=5.0)
    _check(hypos[0][0], [w2, w1, w2, eos], [0.1, 0.9, 0.9, 1.0], lenpen=5.0)
    _check(hypos[0][1], [w1, eos], [0.9, 1.0], lenpen=5.0)
    _check(hypos[1][0], [w1, w2, w1, eos], [0.7, 0.4, 0.4, 1.0], lenpen=5.0)
    _check(hypos[1][1], [w1, w2, eos], [0.7, 0.4, 0.6], lenpen=5.0)
    eos, w1, w2, hypos = _run(search, device, max_len_b=2)  # maxlen
    _check(hypos[0][0], [w1, eos], [0.9, 1.0])
    _check(hypos[0][1], [w2, w2, eos], [0.1, 0.1, 0.6])
    _check(hypos[1][0], [w1, w2, eos], [0.7, 0.4, 0.6])
    _check(hypos[1][1], [w2, w2, eos], [0.3, 0.9, 0.01])


def test_known_answers_host_logic_cpu():
    _assert_scenarios(TorchRefSearch(), "cpu")


def test_ensemble_of_identical_members_reproduces_the_single_model_answers():
    """fairseq/sequence_generator.py:837-939 EnsembleModel: log of the mean member probability — two copies of the scripted
    model must give the known answers again (logsumexp(x, x) - log 2 = x), through separate incremental states per member."""
    d, w1, w2, sample, m1 = scripted_setup()
    _, _, _, _, m2 = scripted_setup()
    gen = SequenceGenerator([m1, m2], d, beam_size=2, search=TorchRefSearch())
    hypos = gen.generate([m1, m2], sample)
    eos = d.eos()
    _check(hypos[0][0], [w1, eos], [0.9, 1.0])
    _check(hypos[0][1], [w2, w1, w2, eos], [0.1, 0.9, 0.9, 1.0])
    _check(hypos[1][0], [w1, w2, w1, eos], [0.7, 0.4, 0.4, 1.0])
    _check(hypos[1][1], [w1, w2, eos], [0.7, 0.4, 0.6])


@pytest.mark.gpu
def test_known_answers_hip_beam_kernels():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _assert_scenarios(HipBeamSearch(), "cuda:0")


@pytest.mark.gpu
def test_beam_kernels_match_torch_search():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    g = torch.Generator().manual_seed(0)
    bsz, beam, V = 5, 4, 503
    ref, hip = TorchRefSearch(), HipBeamSearch()
    for step, only_eos, forbid, ef in ((0, False, True, None), (3, False, False, 1.5), (7, True, False, None), (2, False, False, None)):
        lp = torch.log_softmax(torch.randn(bsz * beam, V, generator=g) * 3, -1)
        lp[1, 7] = float("nan")
        prev = torch.randn(bsz * beam, generator=g)
        a = ref.mask(lp, 1, 3, 2, 0.5, only_eos, forbid, ef)
        b = hip.mask(lp.clone().cuda(), 1, 3, 2, 0.5, only_eos, forbid, ef)
        assert torch.equal(a, b.cpu())
        s1, t1, b1 = ref.step(step, a, prev, bsz, beam)
        s2, t2, b2 = hip.step(step, b, prev.cuda(), bsz, beam)
        assert torch.allclose(s1, s2.cpu(), atol=1e-6)
        fin = torch.isfinite(s1)
        assert torch.equal(t1[fin], t2.cpu()[fin]) and torch.equal(b1[fin], b2.cpu()[fin])
