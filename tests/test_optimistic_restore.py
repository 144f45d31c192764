"""`models.optimistic_restore` (the Flickr30k fine-tune restore, reference models/__init__.py:14-41) against what the REFERENCE function
did to a reference network on two fabricated checkpoints (tests/golden/make_golden.py `restore_cases`, regenerated with --only-restore):
(a) larger-vocabulary checkpoint + an unknown key + a missing key, through a word map with kept rows -> False, (b) same-shape checkpoint
through a permuting word map -> True.  CPU only: the function is host code over `state_dict()` tensors."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

import subgc.models as models

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _case(case):
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        meta = json.load(f)[f"restore_{case}"]
    with np.load(os.path.join(GOLDEN, "restore_out.npz")) as z:
        g = {k: z[k] for k in z.files if k.startswith(case + ".")}
    pick = lambda grp: {k[len(case) + len(grp) + 2:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith(f"{case}.{grp}.")}
    return meta, pick("before"), pick("ckpt"), pick("after"), g[f"{case}.word_map"], bool(g[f"{case}.ok"])


@pytest.mark.parametrize("case", ["a", "b"])
def test_optimistic_restore_equals_the_reference(case, tmp_path, monkeypatch, capsys):
    meta, before, ckpt, after, word_map, ok = _case(case)
    net = models.setup(argparse.Namespace(**dict(meta["opt"], obj_name_path=None, rel_name_path=None)))
    assert set(net.state_dict().keys()) == set(before.keys())
    net.load_state_dict(before)
    (tmp_path / "data").mkdir()
    np.save(tmp_path / "data" / "word_mapping.npy", word_map)
    monkeypatch.chdir(tmp_path)                                  # the reference reads 'data/word_mapping.npy' relative to the cwd; so does the default here
    got = models.optimistic_restore(net, ckpt)
    assert got == ok == meta["returned"]
    own = net.state_dict()
    changed = 0
    for k, b in before.items():
        want = after.get(k, b)                                   # the fixture stores `after` only where the reference changed a tensor
        assert torch.equal(own[k].cpu(), want), k
        changed += k in after
    assert changed > 5 and "embed.0.weight" in after
    said = capsys.readouterr().out
    assert "copy COCO-pre-trained embedding done!" in said
    if case == "a":
        assert "Unexpected key not_in_the_network.weight" in said and "We couldn't find ctx2att.bias" in said
        assert "Network has logit.weight with size torch.Size([31, 16]), ckpt has torch.Size([51, 16])" in said
        # a size-mismatched tensor keeps the network's own values, except the remapped embedding rows
        assert torch.equal(own["logit.weight"].cpu(), before["logit.weight"])
        rows = np.nonzero(word_map != -1)[0]
        assert torch.equal(own["embed.0.weight"].cpu()[rows], ckpt["embed.0.weight"][word_map[rows]])
        keep = np.nonzero(word_map == -1)[0]
        assert torch.equal(own["embed.0.weight"].cpu()[keep], before["embed.0.weight"][keep])


def test_optimistic_restore_explicit_word_map_path(tmp_path):
    meta, before, ckpt, after, word_map, ok = _case("b")
    net = models.setup(argparse.Namespace(**dict(meta["opt"], obj_name_path=None, rel_name_path=None)))
    net.load_state_dict(before)
    path = tmp_path / "wm.npy"
    np.save(path, word_map)
    assert models.optimistic_restore(net, ckpt, word_map_path=str(path)) is True
    assert torch.equal(net.state_dict()["embed.0.weight"].cpu(), after["embed.0.weight"])
