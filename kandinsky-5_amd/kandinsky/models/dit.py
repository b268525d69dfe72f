"""DiffusionTransformer3D — host mirror of the reference class (kandinsky/models/dit.py:82-186).

Same constructor kwargs, same `state_dict` keys, same `forward` signature and
`visual_cond` attribute, so `get_T2V_pipeline`, `generate` and the ComfyUI nodes can use it unchanged.
The arithmetic is NOT here: `forward` hands raw device pointers to the gfx950 engine in libk5.so
(kandinsky-5_amd/csrc/engine.hip) through the C ABI (include/k5.h).  There is no eager fallback: without
the library, or with CPU tensors, `forward` raises.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from .. import _engine as E
from .nn import (FeedForward, Modulation, MultiheadCrossAttention, MultiheadSelfAttentionDec,
                 MultiheadSelfAttentionEnc, NormWeight, OutLayer, TextEmbeddings, TimeEmbeddings,
                 VisualEmbeddings, _NoMath)


class TransformerEncoderBlock(_NoMath):  # reference dit.py:22-44
    def __init__(self, model_dim, time_dim, ff_dim, head_dim):
        super().__init__()
        self.text_modulation = Modulation(time_dim, model_dim, 6)
        self.self_attention = MultiheadSelfAttentionEnc(model_dim, head_dim)
        self.feed_forward = FeedForward(model_dim, ff_dim)


class TransformerDecoderBlock(_NoMath):  # reference dit.py:47-79
    def __init__(self, model_dim, time_dim, ff_dim, head_dim):
        super().__init__()
        self.visual_modulation = Modulation(time_dim, model_dim, 9)
        self.self_attention = MultiheadSelfAttentionDec(model_dim, head_dim)
        self.cross_attention = MultiheadCrossAttention(model_dim, head_dim)
        self.feed_forward = FeedForward(model_dim, ff_dim)


def sp_transport(transport=None):
    """"rccl" | "ipc": explicit argument, else K5_SP_TRANSPORT, else RCCL (the transport of a one-rank-per-GPU node)."""
    import os
    t = (transport or os.environ.get("K5_SP_TRANSPORT", "rccl")).lower()
    if t not in ("rccl", "ipc"):
        raise ValueError(f"unknown sequence-parallel transport {t!r} (rccl | ipc)")
    return t


def _broadcast_ipc_name(kind, is_root, world, group, src):
    """The name of an IPC group's shared-memory control block: made up on the group's rank 0, carried by torch.distributed."""
    import os
    import uuid
    payload = [f"/k5ipc_{kind}_{os.getpid()}_{uuid.uuid4().hex[:12]}" if is_root else None]
    if world > 1:
        import torch.distributed as dist
        dist.broadcast_object_list(payload, src=src, group=group)
    return payload[0]


class DiffusionTransformer3D(nn.Module):
    def __init__(
        self,
        in_visual_dim=4,
        in_text_dim=3584,
        in_text_dim2=768,
        time_dim=512,
        out_visual_dim=4,
        patch_size=(1, 2, 2),
        model_dim=2048,
        ff_dim=5120,
        num_text_blocks=2,
        num_visual_blocks=32,
        axes_dims=(16, 24, 24),
        visual_cond=False,
    ):
        super().__init__()
        head_dim = sum(axes_dims)
        self.in_visual_dim = in_visual_dim
        self.out_visual_dim = out_visual_dim
        self.model_dim = model_dim
        self.patch_size = tuple(patch_size)
        self.visual_cond = visual_cond
        self._cfg = dict(in_visual_dim=in_visual_dim, in_text_dim=in_text_dim, in_text_dim2=in_text_dim2,
                         time_dim=time_dim, out_visual_dim=out_visual_dim, patch_size=tuple(patch_size),
                         model_dim=model_dim, ff_dim=ff_dim, num_text_blocks=num_text_blocks,
                         num_visual_blocks=num_visual_blocks, axes_dims=tuple(axes_dims), visual_cond=bool(visual_cond))

        visual_embed_dim = 2 * in_visual_dim + 1 if visual_cond else in_visual_dim
        self.visual_embed_dim = visual_embed_dim
        self.time_embeddings = TimeEmbeddings(model_dim, time_dim)
        self.text_embeddings = TextEmbeddings(in_text_dim, model_dim)
        self.pooled_text_embeddings = TextEmbeddings(in_text_dim2, time_dim)
        self.visual_embeddings = VisualEmbeddings(visual_embed_dim, model_dim, patch_size)
        self.text_transformer_blocks = nn.ModuleList(
            [TransformerEncoderBlock(model_dim, time_dim, ff_dim, head_dim) for _ in range(num_text_blocks)])
        self.visual_transformer_blocks = nn.ModuleList(
            [TransformerDecoderBlock(model_dim, time_dim, ff_dim, head_dim) for _ in range(num_visual_blocks)])
        self.out_layer = OutLayer(model_dim, time_dim, out_visual_dim, patch_size)

        self._handle = None          # k5_dit*
        self._handle_device = None
        self._keepalive = []
        self._sp = None              # (rank, world) once a communicator lives on the handle
        self._cfg_pair = None        # CFG-parallel branch (0 / 1) once the pair communicator lives on the handle
        self._settings = {"fp8": False, "graph": False, "options": {}}   # re-applied when the engine is rebuilt

    # ---------------------------------------------------------------- engine lifetime
    def _destroy_engine(self, force=False):
        if self._handle is not None:
            if (getattr(self, "_sp", None) is not None or getattr(self, "_cfg_pair", None) is not None) and not force:
                # a rank that silently rebuilt its engine would run the unsharded forward while its peers wait in a collective
                raise RuntimeError("this DiffusionTransformer3D holds a live sequence-parallel communicator: replacing its "
                                   "weights or moving it to another device would desynchronise the ranks")
            E.lib().k5_dit_destroy(self._handle)
        self._handle, self._handle_device, self._sp, self._cfg_pair = None, None, None, None

    def __del__(self):
        try:
            self._destroy_engine(force=True)
        except Exception:
            pass

    def _create_handle(self):
        c = self._cfg
        cc = E.DitConfig(c["in_visual_dim"], c["in_text_dim"], c["in_text_dim2"], c["time_dim"], c["out_visual_dim"],
                         (C.c_int * 3)(*c["patch_size"]), c["model_dim"], c["ff_dim"], c["num_text_blocks"],
                         c["num_visual_blocks"], (C.c_int * 3)(*c["axes_dims"]), int(c["visual_cond"]))
        h = C.c_void_p()
        E.check(E.lib().k5_dit_create(C.byref(cc), C.byref(h)), "k5_dit_create")
        return h

    @staticmethod
    def _load_one(handle, name, t):
        t = t.detach()
        if t.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            t = t.float()
        t = t.contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        E.check(E.lib().k5_dit_load_tensor(handle, name.encode(), t.data_ptr(), E.k5_dtype(t), shape, t.dim()),
                f"k5_dit_load_tensor({name})")

    def _build_engine(self, device):
        """Pack the current parameters into the HIP engine on `device` (once; rebuilt if the
        parameters are replaced or the module is moved)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the Kandinsky-5 HIP engine runs on an MI355X only (device 'cuda:N'); no CPU path")
        self._destroy_engine()
        with torch.cuda.device(device):
            h = self._create_handle()
            for name, t in self.state_dict().items():
                if t.is_meta:
                    E.lib().k5_dit_destroy(h)
                    raise RuntimeError(f"parameter {name} is on the meta device: load a checkpoint first")
                self._load_one(h, name, t)
            E.check(E.lib().k5_dit_finalize(h), "k5_dit_finalize")
        self._handle, self._handle_device = h, device
        self._reapply_settings()

    def _reapply_settings(self):
        """Engine state that lives on the handle (not in the parameters) survives a rebuild."""
        st = self._settings
        if st["fp8"]:
            E.check(E.lib().k5_dit_set_fp8(self._handle, int(st["fp8"])), "k5_dit_set_fp8")
        if st["graph"]:
            E.check(E.lib().k5_dit_set_graph(self._handle, 1), "k5_dit_set_graph")
        for k, v in st["options"].items():
            if k != "emulate_world":
                E.check(E.lib().k5_dit_set_option(self._handle, k.encode(), int(v)), f"k5_dit_set_option({k})")
        if getattr(self, "mag_ratios", None) is not None:   # set_magcache_params() before the weights were loaded
            from ..magcache_utils import _apply
            _apply(self)

    def init_synthetic(self, device, seed=0, std=0.02, qk_gain=1.0, host_rng=False):
        """Random-init weights of this architecture generated tensor by tensor and handed straight to the engine (no 8 GB host
        copy).  Linear ~ N(0,std^2) incl. Modulation (reference zero-inits it, nn.py:158-159, which would make every block an
        identity), norm weights 1, biases N(0,std^2).  Drawn ON DEVICE by default; `host_rng=True` draws every tensor from its own
        CPU generator seeded (seed * 1000003 + index in state_dict order) — the streams a CPU process can reproduce, so that a
        reference run on the host sees the very same weights (bench.py's parity check against tests/golden/dit_fulldepth_c2.*)."""
        device = torch.device(device)
        self._destroy_engine()
        with torch.cuda.device(device):
            h = self._create_handle()
            g = torch.Generator(device="cpu" if host_rng else device)
            for idx, (name, p) in enumerate(self.state_dict().items()):
                g.manual_seed(seed * 1000003 + idx)
                if name.endswith("norm.weight") and len(p.shape) == 1:
                    t = torch.ones(p.shape, device=device)
                    if name.endswith(("query_norm.weight", "key_norm.weight")):
                        t = t * float(qk_gain)   # QK-norm gains of a trained checkpoint are not 1: bench.py --qk-gain
                else:
                    s = std * (2.5 if "modulation" in name else 1.0)
                    if host_rng:
                        t = (torch.randn(p.shape, generator=g) * s).to(device)
                    else:
                        t = torch.randn(p.shape, device=device, generator=g) * s
                self._load_one(h, name, t)
            torch.cuda.synchronize(device)
            E.check(E.lib().k5_dit_finalize(h), "k5_dit_finalize")
        self._handle, self._handle_device = h, device
        self._reapply_settings()
        return self

    def load_state_dict(self, state_dict, strict=True, assign=False):
        out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self._destroy_engine()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if self._handle is not None:
            p = next(self.parameters(), None)
            # the engine owns packed copies of the weights: `.to("cpu")` (the pipeline's offload, t2v_pipeline.py) keeps the
            # handle — the next forward on the same GPU costs nothing; only a move to ANOTHER GPU rebuilds
            if p is not None and p.device.type == "cuda" and p.device != self._handle_device:
                self._destroy_engine()
        return out

    def engine(self, device):
        device = torch.device(device)
        if device.index is None and device.type == "cuda":
            device = torch.device("cuda", torch.cuda.current_device())
        if self._handle is None or self._handle_device != device:
            self._build_engine(device)
        return self._handle

    # ---------------------------------------------------------------- argument marshalling
    def _text_cond(self, text_embed, pooled_text_embed, text_rope_pos, keep):
        if text_embed.dtype != pooled_text_embed.dtype:
            pooled_text_embed = pooled_text_embed.to(text_embed.dtype)
        if text_embed.dtype not in (torch.float32, torch.bfloat16):
            text_embed, pooled_text_embed = text_embed.float(), pooled_text_embed.float()
        text_embed, pooled_text_embed = text_embed.contiguous(), pooled_text_embed.contiguous()
        pos = E.i32_array(torch.as_tensor(text_rope_pos).tolist())
        keep += [text_embed, pooled_text_embed, pos]
        if len(pos) != text_embed.shape[0]:
            raise ValueError("text_rope_pos must have one position per text token")
        return E.TextCond(text_embed.data_ptr(), pooled_text_embed.data_ptr(), E.k5_dtype(text_embed),
                          text_embed.shape[0], pos)

    def _forward_args(self, x_shape, x_ptr, x_channels, text_embed, pooled_text_embed, time, visual_rope_pos,
                      text_rope_pos, scale_factor, sparse_params, keep):
        T, H, W = x_shape
        pt, ph, pw = self.patch_size
        pos = [E.i32_array(torch.as_tensor(p).tolist()) for p in visual_rope_pos]
        if (len(pos[0]), len(pos[1]), len(pos[2])) != (T // pt, H // ph, W // pw):
            raise ValueError("visual_rope_pos does not match the latent shape")
        keep += pos
        a = E.ForwardArgs()
        a.x, a.T, a.H, a.W, a.x_channels = x_ptr, T, H, W, x_channels
        a.cond = self._text_cond(text_embed, pooled_text_embed, text_rope_pos, keep)
        a.time = float(time)
        a.pos_t, a.pos_h, a.pos_w = pos
        a.scale_factor = (C.c_float * 3)(*[float(s) for s in scale_factor])
        if sparse_params is not None:
            a.attention_type = 1
            a.nabla_P = float(sparse_params["P"])
            a.nabla_wT, a.nabla_wH, a.nabla_wW = int(sparse_params["wT"]), int(sparse_params["wH"]), int(sparse_params["wW"])
        return a

    # ---------------------------------------------------------------- reference API
    @torch.no_grad()
    def forward(self, x, text_embed, pooled_text_embed, time, visual_rope_pos, text_rope_pos,
                scale_factor=(1.0, 1.0, 1.0), sparse_params=None):
        """Reference signature dit.py:155-165.  x (T,H,W,C_in) ; returns velocity (T,H,W,out_visual_dim) bf16."""
        if not x.is_cuda:
            raise RuntimeError("DiffusionTransformer3D.forward needs CUDA (HIP) tensors; there is no CPU fallback")
        h = self.engine(x.device)
        x = x.float().contiguous()
        T, H, W, Cx = x.shape
        text_embed, pooled_text_embed = text_embed.to(x.device), pooled_text_embed.to(x.device)
        t_val = float(time.reshape(-1)[0]) if torch.is_tensor(time) else float(time)
        keep = [x]
        a = self._forward_args((T, H, W), x.data_ptr(), Cx, text_embed, pooled_text_embed, t_val, visual_rope_pos,
                               text_rope_pos, scale_factor, sparse_params, keep)
        out = torch.empty(T, H, W, self.out_visual_dim, dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device(x.device):
            E.check(E.lib().k5_dit_forward(h, C.byref(a), out.data_ptr(), E.stream_ptr(x.device)), "k5_dit_forward")
        return out

    @torch.no_grad()
    def sample(self, latent, sigmas, text_embeds, null_text_embeds, visual_rope_pos, text_rope_pos,
               null_text_rope_pos, guidance_weight, scale_factor=(1.0, 1.0, 1.0), sparse_params=None):
        """Whole Euler/CFG loop on device (generation_utils.py:80-129) in one C call.  `latent` fp32
        (T,H,W,in_visual_dim) is updated in place; `sigmas` = the sigma schedule (num_steps+1 floats, host)."""
        if not latent.is_cuda or latent.dtype != torch.float32 or not latent.is_contiguous():
            raise RuntimeError("latent must be a contiguous fp32 CUDA tensor")
        h = self.engine(latent.device)
        dev = latent.device
        T, H, W, _ = latent.shape
        keep = []
        te, pe = text_embeds["text_embeds"].to(dev), text_embeds["pooled_embed"].to(dev)
        s = E.SampleArgs()
        s.fwd = self._forward_args((T, H, W), None, self.in_visual_dim, te, pe, 0.0, visual_rope_pos, text_rope_pos,
                                   scale_factor, sparse_params, keep)
        if abs(guidance_weight - 1.0) > 1e-6:
            s.null_cond = self._text_cond(null_text_embeds["text_embeds"].to(dev), null_text_embeds["pooled_embed"].to(dev),
                                          null_text_rope_pos, keep)
        sig = [float(v) for v in sigmas]
        arr = (C.c_float * len(sig))(*sig)
        s.latent, s.num_steps, s.sigmas, s.guidance_weight = latent.data_ptr(), len(sig) - 1, arr, float(guidance_weight)
        with torch.cuda.device(dev):
            E.check(E.lib().k5_sample(h, C.byref(s), E.stream_ptr(dev)), "k5_sample")
        return latent

    # ---------------------------------------------------------------- multi-GPU
    def enable_sequence_parallel(self, rank, world, device=None, group=None, src=0, transport=None):
        """Token-sharded sequence parallelism (one process per rank; replaces the reference's DTensor plan,
        kandinsky/models/parallelize.py).  transport "rccl" (default): rank 0 creates the ncclUniqueId inside libk5,
        torch.distributed (already initialised by the launcher, kandinsky/utils.py:40-55 contract) only carries its 128 bytes.
        transport "ipc" (or K5_SP_TRANSPORT=ipc): peers read each other's IPC-mapped slots (k5_dit_comm_init_ipc) — no RCCL, and
        several ranks may share one device; torch.distributed carries the name of the group's shared-memory control block."""
        import os
        if self._handle is None:
            if device is None:
                raise RuntimeError("build the engine first (forward / init_synthetic) or pass device=")
            self.engine(device)
        transport = sp_transport(transport)
        if transport == "ipc":
            name = _broadcast_ipc_name("sp", rank == 0, world, group, src)
            with torch.cuda.device(self._handle_device):
                E.check(E.lib().k5_dit_comm_init_ipc(self._handle, name.encode(), int(rank), int(world)), "k5_dit_comm_init_ipc")
            self._sp = (rank, world)
            return self
        lib_path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path = lib_path.encode() if os.path.exists(lib_path) else None
        payload = [None]
        if rank == 0:
            uid = C.create_string_buffer(128)
            E.check(E.lib().k5_comm_unique_id(path, uid), "k5_comm_unique_id")
            payload = [uid.raw]
        if world > 1:
            import torch.distributed as dist
            dist.broadcast_object_list(payload, src=src, group=group)   # src = global rank of the group's rank 0
        with torch.cuda.device(self._handle_device):
            E.check(E.lib().k5_dit_comm_init(self._handle, path, int(rank), int(world), payload[0]), "k5_dit_comm_init")
        self._sp = (rank, world)
        return self

    def sp_schedule(self):
        """What the self-tuning sequence-parallel schedule measured and chose on this handle (dict; {} before the first sharded forward
        of a multi-rank handle): k5_dit_sp_schedule."""
        import json
        if self._handle is None:
            return {}
        buf = C.create_string_buffer(8192)
        E.lib().k5_dit_sp_schedule(self._handle, buf, 8192)
        return json.loads(buf.value.decode() or "{}")

    def enable_loopback(self, group, rank):
        """Tests: this handle becomes rank `rank` of a loopback group (`kandinsky._engine.LoopbackGroup`) — several handles of
        one process on one GPU run the sequence-parallel code path of a multi-GPU job, one host thread per rank."""
        if self._handle is None:
            raise RuntimeError("build the engine first")
        with torch.cuda.device(self._handle_device):
            E.check(E.lib().k5_dit_comm_init_loopback(self._handle, group.handle, int(rank)), "k5_dit_comm_init_loopback")
        self._sp = (rank, group.world)
        self._keepalive.append(group)
        return self

    def enable_cfg_pair(self, branch, group=None, src=0, device=None, transport=None):
        """CFG-parallel inside the engine (k5_dit_cfg_pair_init): this rank runs ONE branch of classifier-free guidance in `sample`
        (0 = conditional, 1 = unconditional) and exchanges the velocity with its partner — `group` = the 2-rank torch.distributed
        group of the pair (it only carries the 128-byte id from `src`, the global rank of branch 0).  Call after
        enable_sequence_parallel (both are collective).  transport as in enable_sequence_parallel."""
        import os
        if self._handle is None:
            if device is None:
                raise RuntimeError("build the engine first (forward / init_synthetic) or pass device=")
            self.engine(device)
        if sp_transport(transport) == "ipc":
            name = _broadcast_ipc_name("pair", branch == 0, 2, group, src)
            with torch.cuda.device(self._handle_device):
                E.check(E.lib().k5_dit_cfg_pair_init_ipc(self._handle, name.encode(), int(branch)), "k5_dit_cfg_pair_init_ipc")
            self._cfg_pair = int(branch)
            return self
        lib_path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path = lib_path.encode() if os.path.exists(lib_path) else None
        payload = [None]
        if branch == 0:
            uid = C.create_string_buffer(128)
            E.check(E.lib().k5_comm_unique_id(path, uid), "k5_comm_unique_id")
            payload = [uid.raw]
        import torch.distributed as dist
        dist.broadcast_object_list(payload, src=src, group=group)
        with torch.cuda.device(self._handle_device):
            E.check(E.lib().k5_dit_cfg_pair_init(self._handle, path, int(branch), payload[0]), "k5_dit_cfg_pair_init")
        self._cfg_pair = int(branch)
        return self

    def enable_cfg_pair_loopback(self, group, branch):
        """Tests: the pair as a loopback group of world 2 (`kandinsky._engine.LoopbackGroup(2)`)."""
        if self._handle is None:
            raise RuntimeError("build the engine first")
        with torch.cuda.device(self._handle_device):
            E.check(E.lib().k5_dit_cfg_pair_init_loopback(self._handle, group.handle, int(branch)), "k5_dit_cfg_pair_init_loopback")
        self._cfg_pair = int(branch)
        self._keepalive.append(group)
        return self

    def set_option(self, name, value):
        """k5_dit_set_option: "attn_mode" (0 = softmax form per head from the data, 1 = online max everywhere),
        "attn_row_offsets" (1 = per-row offsets keep heads with a bound up to 190 on the fixed-offset kernel; default),
        "attn_anchor" (1 = heads beyond that bound keep it too, on offsets anchored at achieved scores; default),
        "attn_fuse_qnorm" (1 = norm_qk + RoPE of the visual queries inside the attention kernel, 2 = under sequence parallelism too;
        default 0, measured neutral),
        "nabla_group_rows" (NABLA on one GPU: 64-query rows per key-tile list / attention workgroup; 0 = by the previous forward's
        kept density (default), 2, 4 — same bits), "sp_nabla_passes" (NABLA under sequence parallelism: 2 = attend the rank's own
        key blocks while the gather is in flight; default 1),
        "sp_slices" (sequence parallelism: exchange K / V^T in this many slices, attend each as it lands; default 1),
        "sp_mode" (sequence parallelism: 0 = K / V^T all-gather (default), 1 = Ulysses all-to-all — token rows traded for heads and back;
        needs heads % ranks == 0 and dense attention, otherwise the gather is used),
        "sp_pass1_tiles", "emulate_world" (timing only)."""
        if self._handle is not None:     # no engine yet: remembered and applied when it is built (_reapply_settings)
            E.check(E.lib().k5_dit_set_option(self._handle, name.encode(), int(value)), f"k5_dit_set_option({name})")
        self._settings["options"][name] = int(value)
        return self

    def reset_softmax_memory(self):
        """Forget which heads the per-row-offset softmax served badly ("attn_pref_reset"): a speed hint that is valid from one step to
        the next of ONE sampling run.  `sample` (k5_sample) does it by itself; a caller stepping `forward` itself calls this per run,
        so that the same seed on the same handle gives the same bits whatever the handle computed before."""
        if self._handle is not None:
            E.check(E.lib().k5_dit_set_option(self._handle, b"attn_pref_reset", 1), "k5_dit_set_option(attn_pref_reset)")
        return self

    def get_option(self, name):
        if self._handle is None:
            if name in self._settings["options"]:
                return self._settings["options"][name]
            raise RuntimeError("get_option before the engine is built: only options set through set_option are known")
        v = C.c_int()
        E.check(E.lib().k5_dit_get_option(self._handle, name.encode(), C.byref(v)), f"k5_dit_get_option({name})")
        return v.value

    def attn_variant_counts(self, reset=False):
        """(fixed-offset, online-max) head launches of the visual self-attention since the last reset."""
        a, b = C.c_longlong(), C.c_longlong()
        E.check(E.lib().k5_dit_attn_variant_counts(self._handle, C.byref(a), C.byref(b), int(reset)))
        return a.value, b.value

    def nabla_block_counts(self):
        """(kept, possible) 64x64 blocks of the NABLA maps computed while profiling was on (bench.py: realised density)."""
        a, b = C.c_longlong(), C.c_longlong()
        E.check(E.lib().k5_dit_nabla_block_counts(self._handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_nabla_tap(self, buf=None):
        """Diagnostics: every NABLA map computed from now on (one-GPU path) is expanded to uint8 [H][nb][nb] into `buf` (a CUDA uint8 tensor the
        caller keeps alive), one after the other; None removes the tap."""
        if buf is not None and (not buf.is_cuda or buf.dtype != torch.uint8 or not buf.is_contiguous()):
            raise ValueError("the NABLA tap needs a contiguous CUDA uint8 tensor")
        self._nabla_tap = buf
        E.check(E.lib().k5_dit_set_nabla_tap(self._handle, E.ptr(buf), 0 if buf is None else buf.numel()), "k5_dit_set_nabla_tap")

    def nabla_tap_count(self):
        a = C.c_longlong()
        E.check(E.lib().k5_dit_nabla_tap_count(self._handle, C.byref(a)), "k5_dit_nabla_tap_count")
        return a.value

    def nabla_executed_blocks(self):
        """64x64 blocks the list-driven attention executed for those maps (union lists x rows per list): kept / executed = union efficiency."""
        a = C.c_longlong()
        E.check(E.lib().k5_dit_nabla_executed_blocks(self._handle, C.byref(a)))
        return a.value

    def set_fp8(self, on=True):
        """opt-in, lossy: linear layers of the visual blocks in W8A8 e4m3 (k5_dit_set_fp8; BASELINE config 5).  `on`: True / 1 = the
        feed-forward GEMMs; a bit mask adds 2 = the q | k | V^T projections and 4 = the out projection of the visual self-attention
        (7 = all three); False / 0 = off."""
        mask = int(on) if not isinstance(on, bool) else (1 if on else 0)
        E.check(E.lib().k5_dit_set_fp8(self._handle, mask), "k5_dit_set_fp8")
        self._settings["fp8"] = mask
        return self

    def set_graph(self, on=True):
        """sample() replays one hipGraph-captured step (k5_dit_set_graph); bit-identical results"""
        E.check(E.lib().k5_dit_set_graph(self._handle, int(on)))
        self._settings["graph"] = bool(on)
        return self

    # ---------------------------------------------------------------- profiling (bench.py roofline)
    def set_profiling(self, on=True):
        E.check(E.lib().k5_dit_set_profiling(self._handle, int(on)))

    def reset_profile(self):
        E.check(E.lib().k5_dit_reset_profile(self._handle))

    def get_profile(self, family):
        ms, n = C.c_double(), C.c_int64()
        E.check(E.lib().k5_dit_get_profile(self._handle, family.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def get_dit(conf):
    """reference dit.py:184-186"""
    conf = dict(conf) if not isinstance(conf, dict) else conf
    return DiffusionTransformer3D(**conf)
