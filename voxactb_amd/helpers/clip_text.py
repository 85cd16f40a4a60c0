"""CLIP text encoder of the act() path on the HIP device.

Reference: peract/helpers/clip/core/clip.py -- `encode_text_with_embeddings` (:426-440) over the text transformer
(:224-254: 12 x [LayerNorm, causal nn.MultiheadAttention, LayerNorm, Linear-QuickGELU-Linear], width 512, 8 heads,
77 tokens), loaded by the agent at build() (qattention_peract_bc_agent.py:324-328) and called once per act() (:661-664);
`tokenize` (:635-649) + `SimpleTokenizer` (simple_tokenizer.py) in front of it.

Neither the RN50 checkpoint nor the BPE vocabulary ships with voxactb_amd (OpenAI assets the reference downloads /
carries): `ClipTextEncoder` takes the checkpoint's state dict (the text half is used, fp16 tensors are widened),
`SimpleTokenizer` takes the path of `bpe_simple_vocab_16e6.txt.gz`.  Arithmetic is fp32 throughout (the reference runs the
checkpoint in fp16 on the GPU: its own outputs carry ~1e-3 of rounding, ours are the fp32 values of the same network).

    enc = ClipTextEncoder(torch.load('clip_rn50_state_dict.pt'), device)
    agent.set_text_encoder(enc.encode_text_with_embeddings)      # or: enc.for_agent()
"""
import gzip
import html
import os

import torch

from .. import ops
from .._lib import VoxactbHipError, call, require_cuda

CONTEXT_LENGTH = 77


# ----------------------------------------------------------------------------------------------------------------------
# tokenizer (simple_tokenizer.py: byte-level BPE, lower-cased, whitespace-collapsed text)
# ----------------------------------------------------------------------------------------------------------------------
def _bytes_to_unicode():
    """printable stand-ins for the 256 byte values: the printable latin-1 ranges map to themselves, the rest to 256 + n."""
    keep = list(range(ord('!'), ord('~') + 1)) + list(range(ord('\xa1'), ord('\xac') + 1)) + list(range(ord('\xae'), ord('\xff') + 1))
    codes = keep[:]
    extra = 0
    for b in range(256):
        if b not in keep:
            keep.append(b)
            codes.append(256 + extra)
            extra += 1
    return {b: chr(c) for b, c in zip(keep, codes)}


class SimpleTokenizer:
    """CLIP's tokenizer over a BPE merge list (`bpe_simple_vocab_16e6.txt.gz`: one header line, then one merge per line;
    the first 49152 - 256 - 2 merges are used).  Vocabulary order: 256 byte symbols, the same with '</w>', the merges,
    '<|startoftext|>', '<|endoftext|>'."""

    def __init__(self, bpe_path=None, merges=None):
        import regex
        if merges is None:
            bpe_path = bpe_path or os.environ.get('VOXACTB_CLIP_BPE')
            if not bpe_path or not os.path.exists(bpe_path):
                raise FileNotFoundError('SimpleTokenizer needs CLIP\'s bpe_simple_vocab_16e6.txt.gz (argument or VOXACTB_CLIP_BPE); '
                                        'the file is an OpenAI asset that voxactb_amd does not ship')
            with gzip.open(bpe_path) as f:
                lines = f.read().decode('utf-8').split('\n')
            merges = [tuple(l.split()) for l in lines[1:49152 - 256 - 2 + 1]]
        self.byte_encoder = _bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + '</w>' for v in vocab] + [''.join(m) for m in merges] + ['<|startoftext|>', '<|endoftext|>']
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.cache = {'<|startoftext|>': '<|startoftext|>', '<|endoftext|>': '<|endoftext|>'}
        self.pat = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                                 regex.IGNORECASE)

    def _bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = list(token[:-1]) + [token[-1] + '</w>']
        while len(word) > 1:
            best, at = None, -1
            for i in range(len(word) - 1):                      # lowest-ranked adjacent pair
                r = self.ranks.get((word[i], word[i + 1]))
                if r is not None and (best is None or r < best):
                    best, at = r, i
            if best is None:
                break
            a, b = word[at], word[at + 1]
            merged, i = [], 0
            while i < len(word):                                  # merge EVERY occurrence of that pair, left to right
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = ' '.join(word)
        self.cache[token] = out
        return out

    @staticmethod
    def _clean(text):
        try:                                   # upstream: ftfy.fix_text first (identity on plain ASCII instructions)
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        text = html.unescape(html.unescape(text)).strip()
        return ' '.join(text.split()).strip().lower()

    def encode(self, text):
        ids = []
        for tok in self.pat.findall(self._clean(text)):
            tok = ''.join(self.byte_encoder[b] for b in tok.encode('utf-8'))
            ids.extend(self.encoder[t] for t in self._bpe(tok).split(' '))
        return ids

    def tokenize(self, texts, context_length=CONTEXT_LENGTH):
        """clip.py:635-649: [sot] + ids + [eot], zero padded; too long raises."""
        if isinstance(texts, str):
            texts = [texts]
        sot, eot = self.encoder['<|startoftext|>'], self.encoder['<|endoftext|>']
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [sot] + self.encode(t) + [eot]
            if len(ids) > context_length:
                raise RuntimeError('Input %s is too long for context length %d' % (t, context_length))
            out[i, :len(ids)] = torch.tensor(ids)
        return out


# ----------------------------------------------------------------------------------------------------------------------
# text transformer
# ----------------------------------------------------------------------------------------------------------------------
class ClipTextEncoder:
    """The text half of a CLIP checkpoint (state-dict keys as in clip.py: `token_embedding.weight`, `positional_embedding`,
    `transformer.resblocks.<i>.{ln_1,attn.in_proj_*,attn.out_proj,ln_2,mlp.c_fc,mlp.c_proj}`, `ln_final`,
    `text_projection`) resident on the HIP device as fp32."""

    def __init__(self, state_dict, device):
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise VoxactbHipError('ClipTextEncoder runs on a HIP device only (got %s)' % dev)
        self.dev = dev
        sd = {k: v for k, v in state_dict.items() if not k.startswith('visual.')}

        def P(name):
            if name not in sd:
                raise KeyError('CLIP state dict has no %r' % name)
            return sd[name].detach().to(device=dev, dtype=torch.float32).contiguous()

        self.tok = P('token_embedding.weight')
        self.pos = P('positional_embedding')
        self.L, self.D = self.pos.shape
        self.H = self.D // 64
        if self.D % 64 or self.L > 128:
            raise VoxactbHipError('text transformer width must be a multiple of 64 and the context <= 128 tokens')
        n_layers = len({k.split('.')[2] for k in sd if k.startswith('transformer.resblocks.')})
        self.layers = []
        for i in range(n_layers):
            pre = 'transformer.resblocks.%d.' % i
            self.layers.append({n: P(pre + n) for n in (
                'ln_1.weight', 'ln_1.bias', 'attn.in_proj_weight', 'attn.in_proj_bias', 'attn.out_proj.weight', 'attn.out_proj.bias',
                'ln_2.weight', 'ln_2.bias', 'mlp.c_fc.weight', 'mlp.c_fc.bias', 'mlp.c_proj.weight', 'mlp.c_proj.bias')})
        self.ln_w, self.ln_b = P('ln_final.weight'), P('ln_final.bias')
        self.proj_t = P('text_projection').t().contiguous()          # [embed_dim, width]: x @ text_projection as a Linear

    def encode_text_with_embeddings(self, text):
        """text [n, 77] (or [77]) token ids -> (sentence features [n, embed_dim], token embeddings [n, 77, width])
        (clip.py:426-440)."""
        text = torch.as_tensor(text)
        if text.dim() == 1:
            text = text[None]
        n, L = text.shape
        if L != self.L:
            raise VoxactbHipError('expected %d tokens per sequence, got %d' % (self.L, L))
        tok = text.to(device=self.dev, dtype=torch.int32).contiguous()
        require_cuda(tok)
        D, rows = self.D, n * L
        prev = ops.PRECISION
        ops.PRECISION = 'fp32'                 # exact fp32 products: a latency path, 77 tokens
        try:
            x = torch.empty((rows, D), dtype=torch.float32, device=self.dev)
            call('vxb_embed_rows_f32', tok.view(-1), self.tok, self.pos, x, rows, L, D, self.tok.shape[0])
            for w in self.layers:
                h, _, _ = ops.layernorm_fwd(x, w['ln_1.weight'], w['ln_1.bias'])
                qkv = ops.linear(h, w['attn.in_proj_weight'], w['attn.in_proj_bias'])
                a = torch.empty((rows, D), dtype=torch.float32, device=self.dev)
                call('vxb_attn_causal_small_f32', qkv, a, n, L, self.H)
                x = ops.linear(a, w['attn.out_proj.weight'], w['attn.out_proj.bias'], residual=x)
                h, _, _ = ops.layernorm_fwd(x, w['ln_2.weight'], w['ln_2.bias'])
                f = ops.linear(h, w['mlp.c_fc.weight'], w['mlp.c_fc.bias'])
                call('vxb_quick_gelu_f32', f, f.numel())
                x = ops.linear(f, w['mlp.c_proj.weight'], w['mlp.c_proj.bias'], residual=x)
            emb, _, _ = ops.layernorm_fwd(x, self.ln_w, self.ln_b)
            # features of the end-of-text token = the highest id of each sequence (:437-438)
            eot = (torch.arange(n, device=self.dev) * L + tok.long().argmax(dim=-1)).to(torch.int32).contiguous()
            xe = torch.empty((n, D), dtype=torch.float32, device=self.dev)
            call('vxb_embed_rows_f32', eot, emb, None, xe, n, 1, D, rows)
            feat = ops.linear(xe, self.proj_t)
        finally:
            ops.PRECISION = prev
        return feat, emb.view(n, L, D)

    def for_agent(self, cache=True):
        """callable for `QAttentionPerActBCAgent.set_text_encoder`: tokens [77] -> (lang_goal_emb [1, E], lang_token_embs [1, 77, W]).
        An episode repeats one instruction at every step (rollout_generator.py:233-244), so with `cache` the encoding of the
        last token sequence is kept and reused while the tokens do not change (one 77-element comparison on the device
        instead of the 12-layer transformer; the reference re-encodes every time, agent :661-664 -- same values)."""
        if not cache:
            return lambda tokens: self.encode_text_with_embeddings(tokens)
        last = {}

        def encode(tokens):
            tok = torch.as_tensor(tokens).to(device=self.dev, dtype=torch.int64).reshape(-1)
            if 'tok' in last and last['tok'].shape == tok.shape and bool(torch.equal(last['tok'], tok)):
                return last['out']
            out = self.encode_text_with_embeddings(tok)
            last['tok'], last['out'] = tok.clone(), out
            return out
        return encode
