"""CLIP tokenizer restatement (voxactb_amd/helpers/clip_text.py; reference helpers/clip/core/simple_tokenizer.py,
clip.py:635-649).  The BPE vocabulary is an OpenAI asset that voxactb_amd does not ship: the merge mechanics are tested on a
small merge list, and -- when VOXACTB_CLIP_BPE points at `bpe_simple_vocab_16e6.txt.gz` -- the token ids of fixture F12, which
the REFERENCE tokenizer produced."""
import os

import numpy as np
import pytest
import torch

from voxactb_amd.helpers.clip_text import SimpleTokenizer, _bytes_to_unicode


def test_byte_table_is_a_bijection_onto_printable_symbols():
    t = _bytes_to_unicode()
    assert len(t) == 256 and len(set(t.values())) == 256
    assert t[ord('a')] == 'a' and t[ord('!')] == '!' and t[ord(' ')] == chr(256 + 32) and t[0] == chr(256)


def test_merges_apply_by_rank_to_every_occurrence():
    merges = [('o', 'p'), ('op', 'e'), ('e', 'n</w>'), ('t', 'h'), ('th', 'e</w>'), ('j', 'a'), ('ja', 'r</w>'), ('op', 'en</w>')]
    tk = SimpleTokenizer(merges=merges)
    base = 512                                   # 256 byte symbols + 256 with '</w>'
    sot, eot = base + len(merges), base + len(merges) + 1
    # 'open': o p e n</w> -> (o,p) rank 0 -> op e n</w> -> (op,e) rank 1 beats (e,n</w>) rank 2 -> ope n</w> (no further merge)
    ids = tk.encode('Open  the JAR')             # lower-cased, whitespace collapsed
    sym = {v: k for k, v in tk.encoder.items()}
    assert [sym[i] for i in ids] == ['ope', 'n</w>', 'the</w>', 'jar</w>']
    out = tk.tokenize(['open the jar', 'the'])
    assert out.shape == (2, 77) and out.dtype == torch.long
    assert out[0, 0] == sot and out[0, 5] == eot and int(out[0, 6:].abs().sum()) == 0
    assert out[1, :3].tolist() == [sot, tk.encoder['the</w>'], eot]
    with pytest.raises(RuntimeError):
        tk.tokenize('the ' * 80)
    # punctuation and digits split as the reference's pattern does: one token per digit, runs of other symbols together
    pieces = tk.pat.findall("don't stop: 42!!")
    assert pieces == ['don', "'t", 'stop', ':', '4', '2', '!!']


def test_missing_vocabulary_is_reported():
    old = os.environ.pop('VOXACTB_CLIP_BPE', None)
    try:
        with pytest.raises(FileNotFoundError):
            SimpleTokenizer()
    finally:
        if old is not None:
            os.environ['VOXACTB_CLIP_BPE'] = old


@pytest.mark.skipif(not os.path.exists(os.environ.get('VOXACTB_CLIP_BPE', '')), reason='CLIP BPE vocabulary not available')
def test_token_ids_equal_the_reference_tokenizer(golden):
    g = golden('f12_clip_text')
    tk = SimpleTokenizer()
    got = tk.tokenize([str(s) for s in g['sentences']])
    assert np.array_equal(got.numpy(), g['tokens'])
