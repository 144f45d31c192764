"""Pin the oracle's beam search (oracle/subgc_oracle.py: beam_search / Oracle.sample_beam) against beams
produced by the reference's CaptionModel.beam_search + AttModel._sample_sentences (tests/golden/make_golden.py):
classical beam search, GNMT / average length penalties, and diverse beam search with 2 and 3 groups."""
import numpy as np
import pytest
import torch

from oracle import subgc_oracle as O
from subgc import synthetic

BEAM_CASES = ["subgc_beam3", "subgc_beam2_wu", "subgc_beam4_div", "subgc_beam6_div3"]


def check_beams(ret, done, ref, atol):
    np.testing.assert_array_equal(ret[3].cpu().numpy(), ref["keep_ind"])
    np.testing.assert_array_equal(ret[0].cpu().numpy(), ref["seq"])
    np.testing.assert_allclose(ret[1].cpu().numpy(), ref["seqLogprobs"], atol=atol, rtol=1e-4)
    got_seq = np.stack([np.stack([b["seq"].numpy() for b in beams]) for beams in done])
    got_lps = np.stack([np.stack([b["logps"].numpy() for b in beams]) for beams in done])
    got_p = np.array([[b["p"] for b in beams] for beams in done])
    np.testing.assert_array_equal(got_seq, ref["done_seq"])
    np.testing.assert_allclose(got_lps, ref["done_logps"], atol=atol, rtol=1e-4)
    np.testing.assert_allclose(got_p, ref["done_p"], atol=atol * 20, rtol=1e-5)


@pytest.mark.parametrize("name", BEAM_CASES)
def test_oracle_beam_search_matches_reference(golden, name):
    g = golden(name)
    orc = O.Oracle(g.opt(), golden("subgc_beam").group("weights"))
    ret = orc.sample_beam(*synthetic.sample_args(g.tensors("inputs")), opt=g.meta["sample_opt"])
    check_beams(ret, ret[4], g.group("out"), atol=2e-5)


def test_beam_fixtures_cover_early_and_late_endings(golden):
    lens = np.concatenate([(golden(n).group("out")["done_seq"] != 0).sum(-1).reshape(-1) for n in BEAM_CASES])
    T = golden(BEAM_CASES[0]).group("out")["seq"].shape[1]
    assert (lens == 0).any() and (lens == T).any() and ((lens > 0) & (lens < T)).any()
