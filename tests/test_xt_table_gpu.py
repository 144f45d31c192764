"""Decode-time x->gates table (AttModel.xt_gates_table, subgc_token_rows_f32): with frozen weights the word-input term
of the attention LSTM (AttModel.py:332 -> :409-411) is a function of the token alone.  The lookup must equal embed + GEMM,
follow weight updates, and leave the decoded tokens of the golden cases unchanged."""
import numpy as np
import pytest
import torch

from subgc import functions as F_
from subgc import ops, synthetic
from test_parity_gpu import DEV, build

pytestmark = pytest.mark.gpu


def test_token_rows_kernel_is_a_plain_lookup():
    g = torch.Generator().manual_seed(3)
    for rows, C, ld in ((50, 4000, 4000), (9488, 192, 200), (7, 37, 37)):
        table = torch.randn(rows, ld, generator=g).to(DEV)[:, :C]
        tok = torch.randint(-2, rows + 3, (33,), generator=g).to(DEV)           # out-of-range tokens are clamped like the embedding kernel
        out = torch.full((33, C), 7.0, device=DEV)
        ops.token_rows(table, tok, out)
        assert torch.equal(out, table[tok.clamp(0, rows - 1)])
    assert ops.token_rows(table, tok[:0], out[:0]).shape == (0, C)


def test_table_equals_embed_plus_gemm_and_tracks_the_weights(golden):
    g = golden("subgc_greedy")
    m = build(g, golden("subgc_train").group("weights"), False)
    R = m.rnn_size
    tab = m.xt_gates_table()
    assert tab.shape == (m.vocab_size + 1, 4 * R) and m.xt_gates_table() is tab    # cached while the weights stand
    want = torch.relu(m.P("embed.0.weight").double()) @ m.P("core.att_lstm.weight_ih")[:, 2 * R:].double().t()
    torch.testing.assert_close(tab.double(), want, atol=1e-5, rtol=1e-5)
    with torch.no_grad():
        m.P("embed.0.weight").mul_(0.5)                                            # e.g. an optimizer step between two evaluations
    tab2 = m.xt_gates_table()
    assert tab2 is not tab
    torch.testing.assert_close(tab2, 0.5 * tab, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("name", ["subgc_greedy", "subgc_greedy_nms55"])
def test_decode_with_and_without_the_table(golden, name, monkeypatch):
    g = golden(name)
    m = build(g, golden("subgc_train").group("weights"), False)
    b = {k: v.to(DEV) for k, v in g.tensors("inputs").items()}
    opt = g.meta["sample_opt"]
    with_tab = m(*synthetic.sample_args(b), opt=opt, mode="sample")
    seen = []
    orig = F_.DecodeState.__init__

    def no_table(self, pr, P, N, want_att, xt_table=None):
        seen.append(xt_table is not None)
        orig(self, pr, P, N, want_att, None)

    monkeypatch.setattr(F_.DecodeState, "__init__", no_table)
    m.__dict__.pop("_graph_cache", None)                                          # the cached hipGraph holds a table-backed state
    without = m(*synthetic.sample_args(b), opt=opt, mode="sample")
    assert seen and all(seen)                                                      # the product path did pass a table
    assert torch.equal(with_tab[0], without[0])
    torch.testing.assert_close(with_tab[1], without[1], atol=1e-4, rtol=1e-4)
    np.testing.assert_array_equal(with_tab[0].cpu().numpy(), g.group("out")["seq"])


def test_cached_decode_state_follows_load_state_dict(golden):
    """The one-image decode replays a captured hipGraph holding weight snapshots: new weights must retire it."""
    g = golden("subgc_greedy")
    w = golden("subgc_train").group("weights")
    m = build(g, w, False)
    b = {k: v.to(DEV) for k, v in g.tensors("inputs").items()}
    opt = g.meta["sample_opt"]
    first = m(*synthetic.sample_args(b), opt=opt, mode="sample")
    w2 = {k: (v * (0.5 if k.startswith(("core.", "embed.")) else 1.0)).astype(v.dtype) for k, v in w.items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w2.items()})
    again = m(*synthetic.sample_args(b), opt=opt, mode="sample")
    fresh = build(g, w2, False)(*synthetic.sample_args(b), opt=opt, mode="sample")
    assert torch.equal(again[0], fresh[0])
    torch.testing.assert_close(again[1], fresh[1], atol=1e-5, rtol=1e-5)
    assert not torch.allclose(first[1], again[1])
