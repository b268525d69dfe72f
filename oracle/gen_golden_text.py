"""TEST INFRASTRUCTURE ONLY.  Golden vectors for the text-embedding handoff.

The real encoders (Qwen2.5-VL-7B, CLIP-L) do not exist offline, so the handoff is pinned on TINY random-init models of the same
HF classes, committed as data under tests/golden/tiny_qwen and tests/golden/tiny_clip (config + weights + tokenizer files):
  * the tokenizers are byte-level BPE vocabularies without merges (every byte a token) plus the chat specials;
  * the reference's own Qwen2_5_VLTextEmbedder.__call__ / ClipTextEmbedder.__call__ / Kandinsky5TextEmbedder.encode
    (/root/reference/kandinsky/models/text_embedders.py:19-31,67-107) are run on them — their __init__ is bypassed because it asks
    for flash-attn 2 and torch.compile, which this image lacks; the models are loaded with attn_implementation="sdpa";
  * the reference's processor is AutoProcessor(use_fast=True), which needs torchvision (absent); for text-only input the processor
    forwards its keyword arguments to the tokenizer, which is what the stand-in below does.
Outputs: tests/golden/text_embed_tiny.safetensors (+ _meta.json).  Run in the build container only: python -m oracle.gen_golden_text
"""
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("K5_REFERENCE", "/root/reference")

TEXTS = [["a cat in a blue hat"],
         ["a red fox jumps over the frozen river at dawn, cinematic", "rain"],
         ["x" * 2000]]                      # longer than max_length: truncation
# byte-level tokens: the video template alone is ~900 tokens, so the Qwen limit is set where short prompts fit (ragged, padded batch)
# and the last case is cut; the crop offsets 129 / 41 are the reference's data and stay as they are
MAX_LEN_QWEN, MAX_LEN_CLIP = 1100, 24


def bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return [chr(c) for c in cs]


def build_tiny_qwen(path):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast, Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    chars = bytes_to_unicode()
    specials = ["<|endoftext|>", "<|im_start|>", "<|im_end|>"]
    vocab = {c: i for i, c in enumerate(chars)}
    for s in specials:
        vocab[s] = len(vocab)
    tok = Tokenizer(models.BPE(vocab=vocab, merges=[]))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="<|endoftext|>", eos_token="<|im_end|>",
                                   additional_special_tokens=["<|im_start|>", "<|im_end|>"], padding_side="right")
    fast.save_pretrained(path)
    cfg = Qwen2_5_VLConfig(
        text_config=dict(vocab_size=len(vocab), hidden_size=96, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                         num_key_value_heads=2, max_position_embeddings=2048, bos_token_id=vocab["<|endoftext|>"],
                         eos_token_id=vocab["<|im_end|>"], pad_token_id=vocab["<|endoftext|>"],
                         rope_scaling={"type": "mrope", "mrope_section": [4, 4, 4]}),
        vision_config=dict(depth=1, hidden_size=32, intermediate_size=32, num_heads=2, out_hidden_size=96, patch_size=14,
                           fullatt_block_indexes=[0]))
    torch.manual_seed(0)
    model = Qwen2_5_VLForConditionalGeneration(cfg).to(torch.bfloat16)
    model.save_pretrained(path)


def build_tiny_clip(path):
    from transformers import CLIPTextConfig, CLIPTextModel
    chars = bytes_to_unicode()
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "vocab.json"), "w") as f:
        json.dump(vocab, f, ensure_ascii=False)
    with open(os.path.join(path, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "CLIPTokenizer", "model_max_length": 77, "bos_token": "<|startoftext|>",
                   "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>", "pad_token": "<|endoftext|>"}, f)
    cfg = CLIPTextConfig(vocab_size=len(vocab), hidden_size=48, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4,
                         max_position_embeddings=77, bos_token_id=vocab["<|startoftext|>"], eos_token_id=vocab["<|endoftext|>"],
                         pad_token_id=vocab["<|endoftext|>"])
    torch.manual_seed(1)
    CLIPTextModel(cfg).save_pretrained(path)


class TextOnlyProcessor:
    """What AutoProcessor does with images=None, videos=None: the keyword arguments go to the tokenizer."""

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer

    def __call__(self, text=None, images=None, videos=None, **kw):
        assert images is None and videos is None
        return self.tokenizer(text, **kw)


def main():
    from safetensors.torch import save_file
    from transformers import AutoTokenizer, CLIPTextModel, CLIPTokenizer, Qwen2_5_VLForConditionalGeneration
    qdir, cdir = os.path.join(GOLD, "tiny_qwen"), os.path.join(GOLD, "tiny_clip")
    build_tiny_qwen(qdir)
    build_tiny_clip(cdir)
    # the reference module, without its package __init__ (needs omegaconf / diffusers)
    for name, sub in (("kandinsky", "/kandinsky"), ("kandinsky.models", "/kandinsky/models")):
        m = types.ModuleType(name)
        m.__path__ = [REF + sub]
        sys.modules[name] = m
    os.environ["TORCH_COMPILE_DISABLE"] = "1"
    torch.cuda.get_device_capability = lambda *a, **k: (0, 0)
    import kandinsky.models.text_embedders as rte
    q = object.__new__(rte.Qwen2_5_VLTextEmbedder)
    q.model = Qwen2_5_VLForConditionalGeneration.from_pretrained(qdir, dtype=torch.bfloat16, attn_implementation="sdpa").eval()
    q.processor = TextOnlyProcessor(AutoTokenizer.from_pretrained(qdir))
    q.max_length = MAX_LEN_QWEN
    c = object.__new__(rte.ClipTextEmbedder)
    c.model = CLIPTextModel.from_pretrained(cdir).eval()
    c.tokenizer = CLIPTokenizer.from_pretrained(cdir)
    c.max_length = MAX_LEN_CLIP
    k = object.__new__(rte.Kandinsky5TextEmbedder)
    k.embedder, k.clip_embedder, k.conf = q, c, None
    out, meta = {}, {"max_length_qwen": MAX_LEN_QWEN, "max_length_clip": MAX_LEN_CLIP, "cases": []}
    for i, texts in enumerate(TEXTS):
        for content in ("video", "image"):
            enc, cu = k.encode(texts, type_of_content=content)
            tag = f"c{i}_{content}"
            out[tag + "_text_embeds"] = enc["text_embeds"].contiguous()           # bf16, as the model returns it
            out[tag + "_pooled_embed"] = enc["pooled_embed"].float().contiguous()
            out[tag + "_cu_seqlens"] = cu.contiguous()
            meta["cases"].append({"tag": tag, "texts": texts, "type_of_content": content, "cu_seqlens": cu.tolist()})
    save_file(out, os.path.join(GOLD, "text_embed_tiny.safetensors"))
    with open(os.path.join(GOLD, "text_embed_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", len(out), "tensors;", [(c["tag"], c["cu_seqlens"]) for c in meta["cases"]])


if __name__ == "__main__":
    main()
