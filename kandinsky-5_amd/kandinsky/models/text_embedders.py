"""Text embedders — host mirror of kandinsky/models/text_embedders.py (reference): Qwen2.5-VL-7B last hidden states
(prompt-template tokens cropped) + CLIP-L pooled embedding, via HF transformers on PyTorch-ROCm.

Outside the accelerated hot path: it runs once per clip and its output — `(L,3584)` bf16 token
embeddings + `(1,768)` pooled — is what the engine consumes.  Differences from the reference: no hard requirement on
flash-attn (`attn_implementation="sdpa"`) or torch.compile, and a text-only processor when the VL image / video processors cannot
be built (no torchvision).  The system-prompt templates and crop offsets are data the
encoder was used with (reference text_embedders.py:36-53), kept in text_prompts.json.
"""
import json
import os

import torch

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "text_prompts.json")) as _f:
    PROMPT_TEMPLATE = json.load(_f)


def _freeze(model):
    for p in model.parameters():
        p.requires_grad = False
    return model


class _TextOnlyProcessor:
    """Qwen2_5_VLProcessor restricted to text: same call signature, tokenizer underneath."""

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer

    def __call__(self, text=None, images=None, videos=None, **kw):
        if images is not None or videos is not None:
            raise ValueError("text-only processor: the image / video processors could not be loaded")
        return self.tokenizer(text, **kw)

    def apply_chat_template(self, *a, **k):
        return self.tokenizer.apply_chat_template(*a, **k)

    def batch_decode(self, *a, **k):
        return self.tokenizer.batch_decode(*a, **k)


class ClipTextEmbedder:
    def __init__(self, conf, device):
        from transformers import CLIPTextModel, CLIPTokenizer
        self.model = _freeze(CLIPTextModel.from_pretrained(conf.checkpoint_path).to(device))
        self.tokenizer = CLIPTokenizer.from_pretrained(conf.checkpoint_path)
        self.max_length = conf.max_length

    @torch.no_grad()
    def __call__(self, texts):
        inputs = self.tokenizer(texts, max_length=self.max_length, truncation=True, add_special_tokens=True,
                                padding="max_length", return_tensors="pt").to(self.model.device)
        return self.model(**inputs)["pooler_output"]


class Qwen2_5_VLTextEmbedder:
    PROMPT_TEMPLATE = PROMPT_TEMPLATE

    def __init__(self, conf, device):
        from transformers import AutoProcessor, Qwen2_5_VLForConditionalGeneration
        self.model = _freeze(Qwen2_5_VLForConditionalGeneration.from_pretrained(
            conf.checkpoint_path, dtype=torch.bfloat16, attn_implementation="sdpa", device_map=device))
        try:
            self.processor = AutoProcessor.from_pretrained(conf.checkpoint_path, use_fast=True)
        except (ImportError, OSError):
            # the VL processor also builds the image / video processors (torchvision, preprocessor_config.json); this class only
            # ever passes images=None, videos=None, for which the processor hands its keyword arguments to the tokenizer
            from transformers import AutoTokenizer
            self.processor = _TextOnlyProcessor(AutoTokenizer.from_pretrained(conf.checkpoint_path))
        self.max_length = conf.max_length

    @torch.no_grad()
    def __call__(self, texts, type_of_content="video"):
        template = "\n".join(self.PROMPT_TEMPLATE["template"][type_of_content])
        crop = self.PROMPT_TEMPLATE["crop_start"][type_of_content]
        inputs = self.processor(text=[template.format(t) for t in texts], images=None, videos=None,
                                max_length=self.max_length + crop, truncation=True, return_tensors="pt",
                                padding=True).to(self.model.device)
        hidden = self.model(input_ids=inputs["input_ids"], return_dict=True,
                            output_hidden_states=True)["hidden_states"][-1][:, crop:]
        mask = inputs["attention_mask"][:, crop:]
        cu = torch.cumsum(mask.sum(1), dim=0)
        cu = torch.cat([torch.zeros_like(cu)[:1], cu]).to(dtype=torch.int32)
        return hidden[mask.bool()], cu


class Kandinsky5TextEmbedder:
    def __init__(self, conf, device="cpu"):
        self.embedder = Qwen2_5_VLTextEmbedder(conf.qwen, device)
        self.clip_embedder = ClipTextEmbedder(conf.clip, device)
        self.conf = conf

    def encode(self, texts, type_of_content="image"):
        text_embeds, cu_seqlens = self.embedder(texts, type_of_content=type_of_content)
        return {"text_embeds": text_embeds, "pooled_embed": self.clip_embedder(texts)}, cu_seqlens

    def expand_prompt(self, prompt, max_new_tokens=256):
        """Prompt beautification with the same chat model (reference t2v_pipeline.py:47-88); the instruction text is
        the pipeline's own and is supplied by the caller's config when present."""
        proc, model = self.embedder.processor, self.embedder.model
        instruction = getattr(self.conf, "expand_instruction", None) or (
            "You are a prompt beautifier that transforms short user video descriptions into rich, detailed English "
            "prompts specifically optimized for video generation models. Rewrite Prompt: \"{prompt}\" to get "
            "high-quality video generation. Answer only with expanded prompt.")
        messages = [{"role": "user", "content": [{"type": "text", "text": instruction.format(prompt=prompt)}]}]
        text = proc.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
        inputs = proc(text=[text], images=None, videos=None, padding=True, return_tensors="pt").to(model.device)
        ids = model.generate(**inputs, max_new_tokens=max_new_tokens)
        trimmed = [o[len(i):] for i, o in zip(inputs.input_ids, ids)]
        return proc.batch_decode(trimmed, skip_special_tokens=True, clean_up_tokenization_spaces=False)[0]

    def to(self, device):
        self.embedder.model = self.embedder.model.to(device)
        self.clip_embedder.model = self.clip_embedder.model.to(device)
        return self


def get_text_embedder(conf, device="cpu"):
    return Kandinsky5TextEmbedder(conf, device)
