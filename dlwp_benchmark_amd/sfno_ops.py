"""SFNO encoder / decoder with the rollout's frame assembly on libdlwpmi's one-launch kernels (csrc/sfno_io.hip).

Reference: the encoder / decoder MLPs, position embedding and big skip of torch_harmonics' SphericalFourierNeuralOperatorNet as
dlwpbench constructs it (src/dlwpbench/models/fno/fno.py:183-200) and the frame bookkeeping of
SFNO2DModule.forward's loop (fno.py:217-259; clean form unet.py:64-111): x_t = cat(constants, prescribed, frame), out = frame + net(x_t).

`encode` / `decode` are two autograd nodes around the block stack.  Besides the token tensors they exchange
  * tok_lp  the gathered input channels of every token as bf16 rows (the decoder's big-skip operand, never differentiated itself),
  * link    an fp32 [T, 32] placeholder whose GRADIENT carries the decoder's big-skip token gradient back to the encoder,
  * alias   the residual frame again, so that the gradient of `out = frame + y` reaches the encoder's backward kernel, which
            writes ONE frame gradient (big skip + residual + encoder path) -- no autograd accumulation on frames.
The weight gradients of all four layers wait for the last lead time of the backward pass and run as one segmented product
(token_ops._weight_grad_segments); the narrow ones land in zero-padded temporaries that are sliced into the parameters' gradients.
bf16 operands + bf16 storage only (lib.set_storage("bf16")); other modes use the GEMM path of dlwpbench/sfno.py.
"""
import ctypes as C

import torch

from . import lib as L
from .token_ops import _BF, _act_dtype, _grad_slot, _weight_grad_segments, _WGRAD_MAX_SEGMENTS

KP = 32          # DLWP_SFNO_IO_KP


class _IoArgs(C.Structure):
    """dlwp_sfno_io_args (include/dlwpmi.h)"""
    _fields_ = [("src", C.c_void_p * 3), ("src_bs", C.c_longlong * 3), ("src_c", C.c_int * 3), ("HW", C.c_int), ("T", C.c_int),
                ("E", C.c_int), ("tokens", C.c_void_p), ("tokens_lp", C.c_void_p), ("tok_lp", C.c_void_p), ("w1_img", C.c_void_p),
                ("w2_img", C.c_void_p), ("bias", C.c_void_p), ("pos", C.c_void_p), ("z", C.c_void_p), ("h", C.c_void_p),
                ("frame", C.c_void_p), ("frame_bs", C.c_longlong), ("frame_c", C.c_int), ("frame_c0", C.c_int),
                ("frame_add", C.c_void_p), ("frame_add_bs", C.c_longlong), ("tok_grad", C.c_void_p)]


def _planes(t):
    """(tensor, device pointer of sample 0, batch stride in floats) of an fp32 [B, c, H, W] tensor whose samples are contiguous blocks."""
    if t.dtype != torch.float32:
        t = t.float()
    if not t[0].is_contiguous():
        t = t.contiguous()
    if not t.is_cuda:
        raise L.DlwpError("libdlwpmi needs CUDA/HIP tensors (no CPU fallback)")
    return t, t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else t[0].numel())


def applies(net, in_chans, out_chans):
    E = net.encoder[0].out_channels
    return (_act_dtype() == _BF and net.encoder[0].weight.is_cuda and net.encoder[0].bias is not None
            and L.load().dlwp_sfno_io_supported(E, in_chans, out_chans) == 1)


def _pass_state(net):
    """Per rollout pass (sht.spectral_weight_scope): the eight weight images, the channels-last position embedding and the list
    that collects the lead times' weight-gradient operands."""
    from . import sht
    scope = sht._wexp_scope
    key = ("sfno_io", id(net))
    if scope is not None and key in scope:
        return scope[key]
    lib = L.load()
    enc1, enc2, dec1, dec2 = net.encoder[0], net.encoder[2], net.decoder[0], net.decoder[2]
    E, cin, cout = enc1.out_channels, enc1.in_channels, dec2.out_channels
    elems = lib.dlwp_sfno_io_image_elems(E)
    imgs = torch.empty(8, elems, device=enc1.weight.device, dtype=_BF)
    L.check(lib.dlwp_sfno_io_pack(L.ptr(enc1.weight.detach().contiguous()), L.ptr(enc2.weight.detach().contiguous()),
                                  L.ptr(dec1.weight.detach().contiguous()), L.ptr(dec2.weight.detach().contiguous()), E, cin, cout,
                                  int(dec1.in_channels > E), L.ptr(imgs), L.stream()))
    pos_cl = None
    if net.pos_embed is not None:
        pos_cl = net.pos_embed.detach()[0].permute(1, 2, 0).contiguous()          # [H, W, E]
    st = {"imgs": imgs, "pos_cl": pos_cl, "gpos_cl": None, "uses": 0, "pending": [], "net": net}
    if scope is not None:
        scope[key] = st
    return st


class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, frame_index, w1, b1, w2, pos, *sources):
        lib = L.load()
        st = _pass_state(net)
        E = net.encoder[0].out_channels
        B, _, H, W = sources[frame_index if frame_index is not None else 0].shape
        HW, T = H * W, B * H * W
        dev = w1.device
        a = _IoArgs()
        keep, cin = [], 0
        for i, s in enumerate(sources):
            t, p, bs = _planes(s)
            keep.append(t)
            a.src[i], a.src_bs[i], a.src_c[i] = p, bs, t.shape[1]
            cin += t.shape[1]
        tok_lp = torch.empty(T, KP, device=dev, dtype=_BF)
        z = torch.empty(T, E, device=dev, dtype=_BF)
        h = torch.empty(T, E, device=dev, dtype=_BF)
        t0 = torch.empty(B, H, W, E, device=dev)
        a.HW, a.T, a.E = HW, T, E
        a.tokens, a.tok_lp, a.z, a.h = L.ptr(t0), L.ptr(tok_lp), L.ptr(z), L.ptr(h)
        a.w1_img, a.w2_img = L.ptr(st["imgs"][0]), L.ptr(st["imgs"][1])
        a.bias, a.pos = L.ptr(b1), L.ptr(st["pos_cl"])
        L.check(lib.dlwp_sfno_encode_fwd(C.byref(a), L.stream()))
        link = torch.empty(T, KP, device=dev)                  # its gradient = the decoder's big-skip token gradient
        rec = {"tok_lp": tok_lp, "h_e": h}
        link._dlwp_rec = rec
        if any(ctx.needs_input_grad):
            st["uses"] += 1
        ctx.st, ctx.rec, ctx.z, ctx.dims = st, rec, z, (B, H, W, E, cin)
        ctx.frame_index = frame_index
        ctx.c0 = sum(s.shape[1] for s in sources[:frame_index]) if frame_index is not None else 0
        ctx.cf = sources[frame_index].shape[1] if frame_index is not None else 0
        ctx.params = (w1, b1, w2, pos)
        ctx.mark_non_differentiable(tok_lp)
        ctx.set_materialize_grads(False)          # an unused link / alias output arrives as None instead of a zero-filled tensor
        alias = sources[frame_index].view_as(sources[frame_index]) if frame_index is not None else None
        return t0, tok_lp, link, alias

    @staticmethod
    def backward(ctx, g_t0, _g_tok_lp, g_link, g_alias):
        lib = L.load()
        st, rec = ctx.st, ctx.rec
        B, H, W, E, cin = ctx.dims
        HW, T = H * W, B * H * W
        if g_t0 is None:                          # (the tokens were not used: only the link / alias carry a gradient)
            g_t0 = torch.zeros(B, H, W, E, device=ctx.z.device)
        g = g_t0.contiguous().float()
        dev = g.device
        g_lp = torch.empty(T, E, device=dev, dtype=_BF)
        gh = torch.empty(T, E, device=dev, dtype=_BF)
        a = _IoArgs()
        a.HW, a.T, a.E = HW, T, E
        a.tokens, a.tokens_lp, a.z, a.h = L.ptr(g), L.ptr(g_lp), L.ptr(ctx.z), L.ptr(gh)
        a.w1_img, a.w2_img = L.ptr(st["imgs"][2]), L.ptr(st["imgs"][3])
        keep = []
        if g_link is not None:
            g_link = g_link.contiguous().float()
            a.tok_grad = L.ptr(g_link)
        g_frame = None
        nsrc = len(ctx.needs_input_grad) - 6
        if ctx.frame_index is not None and ctx.needs_input_grad[6 + ctx.frame_index]:
            g_frame = torch.empty(B, ctx.cf, H, W, device=dev)
            a.frame, a.frame_bs, a.frame_c, a.frame_c0 = L.ptr(g_frame), ctx.cf * HW, ctx.cf, ctx.c0
            if g_alias is not None:
                t, p, bs = _planes(g_alias)
                keep.append(t)
                a.frame_add, a.frame_add_bs = p, bs
        L.check(lib.dlwp_sfno_encode_bwd(C.byref(a), L.stream()))
        # position embedding: batch sum of g, accumulated channels-last over the lead times of the pass
        if st["pos_cl"] is not None:
            first = st["gpos_cl"] is None          # the first lead time of the pass writes the sums, the later ones add to them
            if first:
                st["gpos_cl"] = torch.empty(HW * E, device=dev)
            L.check(lib.dlwp_colsum_ex(L.ptr(g), L.ptr(st["gpos_cl"]), B, HW * E, int(first), L.stream()))
        rec.update(g_lp=g_lp, gh_e=gh)
        st["pending"].append(rec)
        st["uses"] -= 1
        grads = [None] * 6 + [None] * nsrc
        if ctx.frame_index is not None:
            grads[6 + ctx.frame_index] = g_frame
        if st["uses"] > 0 and len(st["pending"]) < _WGRAD_MAX_SEGMENTS:
            return tuple(grads)
        w = _flush_weight_grads(st)
        grads[2:6] = [w.get("enc_w1"), w.get("enc_b1"), w.get("enc_w2"), w.get("pos")]
        return tuple(grads)


class _Add2dDesc(C.Structure):
    """dlwp_add2d_desc (include/dlwpmi.h)"""
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("dst_ld", C.c_longlong), ("src_rs", C.c_longlong), ("src_cs", C.c_longlong),
                ("rows", C.c_int), ("cols", C.c_int)]


def _flush_weight_grads(st):
    """One segmented product launch for the encoder's and the decoder's weight gradients over every pending lead time; the
    decoder's parameters receive theirs through their gradient slots or, without slots, through `st["dec_grads"]` (returned by the
    decoder node of the same lead time is impossible -- it has already run -- so slot-less decoder parameters are accumulated by
    hand into .grad)."""
    net = st["net"]
    segs, st["pending"] = st["pending"], []
    enc1, enc2, dec1, dec2 = net.encoder[0], net.encoder[2], net.decoder[0], net.decoder[2]
    E, cin, cout = enc1.out_channels, enc1.in_channels, dec2.out_channels
    dev = enc1.weight.device
    col = lambda k: [s[k] for s in segs]
    has_dec = all("gh_d" in s for s in segs)
    def temp(*shape):
        t = torch.empty(*shape, device=dev)
        t._dlwp_overwrite = True              # _weight_grad_segments writes the sum instead of adding to it
        return t
    t_e1 = temp(E, KP)
    layers = [(col("g_lp"), col("h_e"), _grad_slot(enc2.weight), None, False, enc2.weight.shape),
              (col("gh_e"), col("tok_lp"), t_e1, _grad_slot(enc1.bias), True, (E, KP))]
    if has_dec:
        t_da, t_db, t_d2 = temp(E, E), temp(E, KP), temp(KP, E)
        layers += [(col("gh_d"), col("t_lp"), t_da, _grad_slot(dec1.bias) if dec1.bias is not None else None, dec1.bias is not None, (E, E)),
                   (col("gh_d"), col("tok_lp"), t_db, None, False, (E, KP)),
                   (col("g_lp_d"), col("h_d"), t_d2, None, False, (KP, E))]
    outs = _weight_grad_segments(layers)
    res = {}

    def give(param, grad):
        """into the gradient slot when the engine preallocated one, else returned / accumulated as autograd would"""
        slot = _grad_slot(param)
        if slot is not None:
            slot.add_(grad.reshape(slot.shape))
            return None
        return grad.reshape(param.shape)
    res["enc_w2"] = outs[0][0]
    res["enc_b1"] = outs[1][1]
    slots = [_grad_slot(p) for p in (enc1.weight, dec1.weight, dec2.weight)] + ([_grad_slot(net.pos_embed)] if st["gpos_cl"] is not None else [])
    if has_dec and all(sl is not None for sl in slots) and (dec1.bias is None or outs[2][1] is None):
        # every parameter has its gradient slot (the engine's flat buffer): the padded / split / channels-last temporaries join their
        # slots in ONE launch (dlwp_add2d_many) instead of four add_ launches and a cat
        descs, n = (_Add2dDesc * 8)(), 0
        def put(dst, dst_ld, src, rs, cs, rows, cols):
            nonlocal n
            descs[n] = _Add2dDesc(dst, src, dst_ld, rs, cs, rows, cols)
            n += 1
        din = dec1.in_channels
        put(L.ptr(slots[0]), cin, L.ptr(t_e1), KP, 1, E, cin)
        put(L.ptr(slots[1]), din, L.ptr(t_da), E, 1, E, E)
        if din > E:
            put(L.ptr(slots[1]) + 4 * E, din, L.ptr(t_db), KP, 1, E, din - E)
        put(L.ptr(slots[2]), E, L.ptr(t_d2), E, 1, cout, E)
        if st["gpos_cl"] is not None:
            HWp = net.pos_embed.shape[2] * net.pos_embed.shape[3]
            put(L.ptr(slots[3]), HWp, L.ptr(st["gpos_cl"]), 1, E, E, HWp)           # [HW][E] -> [E][HW]
            st["gpos_cl"] = None
        L.check(L.load().dlwp_add2d_many(C.cast(descs, C.c_void_p), n, L.stream()))
        res["enc_w1"] = res["pos"] = None
        return res
    res["enc_w1"] = give(enc1.weight, t_e1[:, :cin])
    if has_dec:
        gwd = torch.cat([t_da, t_db[:, :dec1.in_channels - E]], dim=1) if dec1.in_channels > E else t_da
        for p, gr in ((dec1.weight, gwd), (dec2.weight, t_d2[:cout])):
            r = give(p, gr)
            if r is not None:                      # no slot: the decoder's node has returned already -- accumulate like autograd
                p.grad = r.clone() if p.grad is None else p.grad + r
        if dec1.bias is not None and outs[2][1] is not None:
            p = dec1.bias
            p.grad = outs[2][1] if p.grad is None else p.grad + outs[2][1]
    if st["gpos_cl"] is not None:
        H, W = net.pos_embed.shape[2], net.pos_embed.shape[3]
        gp = st["gpos_cl"].view(H, W, E).permute(2, 0, 1).unsqueeze(0)
        res["pos"] = give(net.pos_embed, gp)
        st["gpos_cl"] = None
    return res


class _DecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, t, tok_lp, link, alias, w1, b1, w2):
        lib = L.load()
        st = _pass_state(net)
        E = net.encoder[0].out_channels
        cout = net.decoder[2].out_channels
        B, H, W, _ = t.shape
        HW, T = H * W, B * H * W
        dev = t.device
        t2 = t.contiguous().float()
        t_lp = torch.empty(T, E, device=dev, dtype=_BF)
        z = torch.empty(T, E, device=dev, dtype=_BF)
        h = torch.empty(T, E, device=dev, dtype=_BF)
        out = torch.empty(B, cout, H, W, device=dev)
        a = _IoArgs()
        a.HW, a.T, a.E = HW, T, E
        a.tokens, a.tokens_lp, a.tok_lp, a.z, a.h = L.ptr(t2), L.ptr(t_lp), L.ptr(tok_lp), L.ptr(z), L.ptr(h)
        a.w1_img, a.w2_img = L.ptr(st["imgs"][4]), L.ptr(st["imgs"][5])
        a.bias = L.ptr(b1)
        a.frame, a.frame_bs, a.frame_c = L.ptr(out), cout * HW, cout
        keep = None
        if alias is not None:
            keep, p, bs = _planes(alias)
            a.frame_add, a.frame_add_bs = p, bs
        L.check(lib.dlwp_sfno_decode_fwd(C.byref(a), L.stream()))
        rec = getattr(link, "_dlwp_rec", None)
        if rec is None:
            raise L.DlwpError("sfno decode: `link` must come from sfno_ops.encode of the same network call")
        rec.update(t_lp=t_lp, h_d=h)
        ctx.st, ctx.rec, ctx.z, ctx.dims, ctx.has_alias = st, rec, z, (B, H, W, E, cout), alias is not None
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = L.load()
        st, rec = ctx.st, ctx.rec
        B, H, W, E, cout = ctx.dims
        HW, T = H * W, B * H * W
        go, p, bs = _planes(g_out)
        dev = go.device
        g_lp = torch.empty(T, KP, device=dev, dtype=_BF)
        gh = torch.empty(T, E, device=dev, dtype=_BF)
        g_t = torch.empty(B, H, W, E, device=dev)
        g_tok = torch.empty(T, KP, device=dev)
        a = _IoArgs()
        a.src[0], a.src_bs[0], a.src_c[0] = p, bs, cout
        a.HW, a.T, a.E = HW, T, E
        a.tokens, a.tok_lp, a.z, a.h, a.tok_grad = L.ptr(g_t), L.ptr(g_lp), L.ptr(ctx.z), L.ptr(gh), L.ptr(g_tok)
        a.w1_img, a.w2_img = L.ptr(st["imgs"][6]), L.ptr(st["imgs"][7])
        L.check(lib.dlwp_sfno_decode_bwd(C.byref(a), L.stream()))
        rec.update(g_lp_d=g_lp, gh_d=gh)
        # weight gradients: with the encoder's, at the end of the pass (_flush_weight_grads)
        return None, g_t, None, g_tok, (g_out if ctx.has_alias else None), None, None, None


def encode(net, sources, frame_index):
    """sources: the plane groups of the network input in channel order ([B, c_i, H, W] each, at most three); frame_index: which of
    them is the differentiable frame that also takes the decoder's residual (None: no residual, nothing differentiable).
    -> (tokens [B, H, W, E], tok_lp, link, alias)"""
    enc1, enc2 = net.encoder[0], net.encoder[2]
    return _EncodeFn.apply(net, frame_index, enc1.weight, enc1.bias, enc2.weight, net.pos_embed, *sources)


def decode(net, t, tok_lp, link, alias):
    """-> out [B, Cout, H, W] = (alias +) decoder([t | network input])"""
    dec1, dec2 = net.decoder[0], net.decoder[2]
    return _DecodeFn.apply(net, t, tok_lp, link, alias, dec1.weight, dec1.bias, dec2.weight)
