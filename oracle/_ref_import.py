"""TEST INFRASTRUCTURE ONLY — never imported by the product path.

Imports the *reference* Kandinsky-5 python (read-only, /root/reference) on CPU under the
six monkey-patches of SURVEY.md Appendix C so that golden vectors can be generated from
the reference's own arithmetic.  Runs ONLY in the build container (the reference tree does
not exist on the GPU box); its outputs travel as data fixtures under tests/golden/.

Patches (all local to the generating process):
  1. TORCH_COMPILE_DISABLE=1                 every @torch.compile is a no-op
  2. torch.cuda.get_device_capability -> (0,0)   (reference nn.py:9 calls it at import)
  3. bare `kandinsky`/`kandinsky.models` module objects (skip kandinsky/__init__.py, which
     needs omegaconf/diffusers/hf_hub)
  4. FA  := SDPA with (B,S,H,D)<->(B,H,S,D) transposes (flash-attn is not installed)
  5. torch.bfloat16 := torch.float32         fp32 oracle mode (neutralises nn.py:28,33,40)
  6. torch.Generator(device="cuda") -> CPU generator (generation_utils.py:97)
  + flex_attention := block-masked SDPA (eager flex ignores the NABLA mask)
"""
import os
import sys
import types

os.environ["TORCH_COMPILE_DISABLE"] = "1"
sys.dont_write_bytecode = True

import torch
import torch.nn.functional as F

REF = os.environ.get("K5_REFERENCE", "/root/reference")


class RefModules:
    pass


def import_reference():
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    torch.cuda.get_device_capability = lambda *a, **k: (0, 0)
    for name, sub in (("kandinsky", "/kandinsky"), ("kandinsky.models", "/kandinsky/models")):
        m = types.ModuleType(name)
        m.__path__ = [REF + sub]
        sys.modules[name] = m
    import kandinsky.models.nn as knn
    import kandinsky.models.dit as kdit
    import kandinsky.models.utils as kutils

    def _fa(q, k, v):
        return F.scaled_dot_product_attention(
            q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
        ).transpose(1, 2)

    def _flex(q, k, v, block_mask=None):
        dense = block_mask.to_dense().bool()
        dense = dense.repeat_interleave(64, -2).repeat_interleave(64, -1)
        return F.scaled_dot_product_attention(q, k, v, attn_mask=dense)

    knn.FA = _fa
    knn.flex_attention = _flex
    torch.bfloat16 = torch.float32  # fp32 oracle mode
    _G = torch.Generator
    torch.Generator = lambda device=None: _G("cpu")
    import kandinsky.generation_utils as kgen

    r = RefModules()
    r.nn, r.dit, r.utils, r.gen = knn, kdit, kutils, kgen
    return r
