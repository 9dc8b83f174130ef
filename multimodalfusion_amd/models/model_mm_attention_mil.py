"""Multimodal (radiology + pathology + omic) attention-MIL with concat / tensor fusion; drop-in for
models/model_mm_attention_mil.py of the reference (ctor signatures :19-23 / :118-121, forward :128-200,
state_dict keys of Appendix B).

The reference class cannot be constructed or run as shipped.  What this module does
about each defect -- the signature and the mathematics are kept, nothing else is changed:
  * `gate_omic=` is accepted by the subclass and not forwarded (the reference forwards it to a base ctor
    without that parameter -> TypeError, :124);
  * the fused width uses size_WSI (the reference names an undefined `size_path`, :83);
  * `genomic_features` may be [G] (what the forward expects, :165) or [1 x G] (what the collate delivers);
  * radio_fusion='tensor' raises NotImplementedError (the reference calls an attribute that is never
    defined, :141);
  * return_features=True returns the fused embedding (the reference raises NameError, :196-198).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..utils.utils import initialize_weights
from .model_modules import Attn_Net, Attn_Net_Gated, SNN_Block, XlinearFusion, amil_stack, snn_stack


class MM_MIL_Attention_fc(nn.Module):
    def __init__(self, input_dim: int = 80, radio_fusion="concat", fusion="tensor", gate=True, gate_path=True,
                 gate_radio=True, dropout=True, model_size_radio: str = "small", model_size_wsi: str = "small",
                 model_size_omic: str = "small", n_classes=4, modalities=["T1", "T2", "T1Gd", "FLAIR"],
                 mode="radio_path_omic"):
        super().__init__()
        self.radio_fusion = radio_fusion
        self.fusion = fusion
        self.n_classes = n_classes
        self.size_dict_radio = {"small": [1024, 256, 256], "big": [1024, 256, 384]}
        self.size_dict_WSI = {"small": [1024, 256, 256], "big": [1024, 256, 384]}
        self.size_dict_omic = {"small": [256, 256], "big": [1024, 256]}
        self.modalities = modalities
        self.mode = mode

        size_omic = self.size_dict_omic[model_size_omic]
        fc_omic = [SNN_Block(dim1=input_dim, dim2=size_omic[0])]
        for i, _ in enumerate(size_omic[1:]):
            fc_omic.append(SNN_Block(dim1=size_omic[i], dim2=size_omic[i + 1], dropout=0.25))
        self.fc_omic = nn.Sequential(*fc_omic)

        size_radio = self.size_dict_radio[model_size_radio]
        fc_radio = [nn.Linear(size_radio[0], size_radio[1]), nn.ReLU(), nn.Dropout(0.25)]
        if gate_radio:
            att = Attn_Net_Gated(L=size_radio[1], D=size_radio[2], dropout=dropout, n_classes=1)
        else:
            att = Attn_Net(L=size_radio[1], D=size_radio[2], dropout=dropout, n_classes=1)
        fc_radio.append(att)
        self.attention_net_radio = nn.Sequential(*fc_radio)

        if self.radio_fusion == "tensor":
            raise NotImplementedError("radio_fusion='tensor' is unusable in the reference and not provided")
        elif self.radio_fusion == "concat":
            self.reduce_dim = nn.Linear(size_radio[0] * len(self.modalities), size_radio[0])

        size_WSI = self.size_dict_WSI[model_size_wsi]
        fc_WSI = [nn.Linear(size_WSI[0], size_WSI[1]), nn.ReLU(), nn.Dropout(0.25)]
        if gate_path:
            att = Attn_Net_Gated(L=size_WSI[1], D=size_WSI[2], dropout=dropout, n_classes=1)
        else:
            att = Attn_Net(L=size_WSI[1], D=size_WSI[2], dropout=dropout, n_classes=1)
        fc_WSI.append(att)
        self.attention_net_WSI = nn.Sequential(*fc_WSI)

        classifier_size = 0
        n_modalities = 0
        if "radio" in mode:
            classifier_size += size_radio[1]
            n_modalities += 1
        if "path" in mode:
            classifier_size += size_WSI[1]
            n_modalities += 1
        if "omic" in mode:
            classifier_size += size_omic[1]
            n_modalities += 1

        if self.fusion == "tensor":
            self.mm = XlinearFusion(dim=256, scale_dim=16, mmhid1=512, mmhid2=512, num_modalities=n_modalities,
                                    gate=gate, skip=1)
            self.classifier = nn.Sequential(*[nn.Linear(512, 256), nn.ReLU(), nn.Dropout(0.25),
                                              nn.Linear(256, n_classes)])
        elif self.fusion == "concat":
            self.classifier = nn.Linear(classifier_size, n_classes)
        initialize_weights(self)

    def relocate(self):
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.fc_omic = self.fc_omic.to(device)
        self.attention_net_radio = self.attention_net_radio.to(device)
        self.attention_net_WSI = self.attention_net_WSI.to(device)
        self.classifier = self.classifier.to(device)
        if self.fusion == "tensor":
            self.mm = self.mm.to(device)
        if self.radio_fusion == "concat":
            self.reduce_dim = self.reduce_dim.to(device)

    def forward(self, h, return_features=False, attention_only=False):
        pass


class MM_MIL_Attention_fc_surv(MM_MIL_Attention_fc):
    def __init__(self, input_dim: int = 80, radio_fusion: str = "concat", fusion: str = "tensor", gate=True,
                 gate_path=True, gate_omic=True, gate_radio=True, model_size_radio="small",
                 model_size_wsi: str = "small", model_size_omic="small", dropout=False, n_classes=4,
                 mode="radio_path_omic"):
        super().__init__(input_dim=input_dim, radio_fusion=radio_fusion, fusion=fusion, gate=gate,
                         gate_path=gate_path, gate_radio=gate_radio, model_size_radio="small",
                         model_size_wsi=model_size_wsi, model_size_omic=model_size_omic, dropout=dropout,
                         n_classes=n_classes, mode=mode)

    def _side_stream(self, device):
        """A second HIP stream for the small branches (radio stack, omic SNN): they are independent of the pathology
        stack until the fusion, and its big kernels leave CUs idle (224 of 256 at 50k instances), so the small
        kernels run beside them instead of after them.  Autograd replays each branch on the stream it ran on.
        One per main stream, chosen by measurement (streams.stream_beside): a pool stream that shares the main stream's
        hardware queue would run the branches after the stack, not beside it."""
        from ..streams import stream_beside
        cur = torch.cuda.current_stream(device)
        pool = self.__dict__.setdefault("_mmf_side", {})        # not a parameter / buffer: stays out of state_dict
        st = pool.get(cur.cuda_stream)
        if st is None or st.device != device:
            st = pool[cur.cuda_stream] = stream_beside([cur], device)
        return st

    def _fork_ok(self, path_x, min_units=120_000):
        # worth it only while the step is GPU-bound, i.e. the pathology stack runs for longer than the host needs to
        # issue the step (~0.9 ms): >= 30k fp32 instances / >= 120k bf16 instances (measured: 50k fp32 1.30 -> 1.07 ms;
        # 100k bf16 is host-bound and the extra stream calls cost 0.05 ms)
        return (getattr(self, "mmf_side_stream", True)          # set False on an instance to keep everything on one stream
                and path_x is not None and path_x.is_cuda and ("radio" in self.mode or "omic" in self.mode)
                and path_x.shape[0] * (1 if path_x.dtype == torch.bfloat16 else 4) >= min_units)

    def _concat_order(self):
        has = lambda k: k in self.mode
        if has("radio") and has("path") and not has("omic"):
            return ["radio", "path"]
        if has("radio") and has("omic") and not has("path"):
            return ["radio", "omic"]
        if has("omic") and has("path") and not has("radio"):
            return ["omic", "path"]
        return ["radio", "path", "omic"]

    def nll_step(self, label, c, alpha=0.0, loss_scale=1.0, grad_out=None, accumulate=None, **kwargs):
        """Extension of the reference surface (the training-loop mirror uses it, utils/core_utils.py): the whole training
        step of one patient (fusion='concat'; fusion='tensor' in the heads' configuration) -- what `hazards, S, Y_hat, A_raw = model(**kwargs)`,
        `loss = NLLSurvLoss(alpha)(hazards=hazards, S=S, Y=label, c=c)`, `(loss * loss_scale).backward()` compute together
        (models/model_mm_attention_mil.py:128-200 + utils/loss_utils.py:22-39 + autograd), with the same dropout draws --
        as a fixed sequence of C-ABI calls with no autograd graph: the branches' forward calls (radio and omic on the side
        stream beside the pathology stack), ONE launch for classifier + hazards + loss + their backward
        (mmf_surv_head_nll_step; the branches write their embeddings side by side, so the concatenation is never a
        launch; with the tensor fusion the XlinearFusion block and classifier[0] run in front of it), the branches'
        backward calls.  Gradients are ADDED to the parameters' .grad (a parameter whose .grad is
        None receives the fresh buffer, as autograd does) -- or to `grad_out`, tensors in self.parameters() order,
        overwritten unless `accumulate`.  Returns (hazards, S, Y_hat, A_raw dict, loss, risk), detached."""
        from ..ops import AmilPoolFn, HandCtx, LinearCatFn, _dense_bwd_raw, _dense_fwd_raw
        if self.fusion == "tensor" and not (self.mm.skip and len(self._concat_order()) * self.mm.reduce[0][0][0].weight.shape[0] <= 384):
            raise NotImplementedError("nll_step covers the XlinearFusion configuration the heads use (skip, one patient)")
        params = list(self.parameters())
        if any(not p.requires_grad for p in params):
            raise RuntimeError("nll_step needs every parameter to require grad")
        tr = self.training
        order = self._concat_order()
        width = {"radio": self.attention_net_radio[0].out_features, "path": self.attention_net_WSI[0].out_features,
                 "omic": self.fc_omic[-1][0].out_features}
        off, o = {}, 0
        for k in order:
            off[k] = o
            o += width[k]
        F = o
        path_x = kwargs.get("path_features") if "path" in order else None
        dev = (path_x if path_x is not None else kwargs[self.modalities[0]] if "radio" in order
               else kwargs["genomic_features"]).device
        # the side stream costs this step ~0.12 ms of host time (stream switches, four cross-stream waits) and pays from
        # 20k fp32 rows on (tools/r4_mm_fork.sh: 20k 0.60 -> 0.54 ms, 50k 1.02 -> 0.89; 10k 0.43 -> 0.45; bf16 100k 0.59 -> 0.56)
        fork = self._fork_ok(path_x, getattr(self, "mmf_fork_min_one_call", 80_000))
        grads = {}                                   # parameter -> gradient tensor of this step

        def stack_args(seq):
            lin, att = seq[0], seq[3]
            gated = isinstance(att, Attn_Net_Gated)
            Wa, ba, Wb, bb, Wc, bc = att.stack_params()
            p_h = seq[2].p if tr else 0.0
            p_att = 0.25 if (tr and att.att_dropout) else 0.0
            seed = ops.next_dropout_seed() if tr else 0
            return (lin.weight, lin.bias, Wa, ba, Wb, bb, Wc, bc), (gated, p_h, p_att, seed)

        def stack_backward(ctx, ps, g):
            out = AmilPoolFn.backward(ctx, g, None)
            for p, gr in zip(ps, out[1:9]):
                if p is not None:
                    grads[p] = gr
            return out[0]

        with torch.no_grad():
            feat = torch.empty((1, F), dtype=torch.float32, device=dev)
            slot = lambda k: feat[:, off[k]:off[k] + width[k]]
            if fork:
                cur = torch.cuda.current_stream(dev)
                side = self._side_stream(dev)
                side.wait_stream(cur)
                branch = lambda: torch.cuda.stream(side)
            else:
                import contextlib
                branch = contextlib.nullcontext
            A_raw = {}
            # ---- forward: python order (and with it the dropout-seed order) radio, path, omic as in forward()
            if "radio" in order:
                with branch():
                    xs = [kwargs[m] for m in self.modalities]
                    ctx_cat = None
                    if len(xs) > 1:
                        ctx_cat = HandCtx((True, True) + (False,) * len(xs))
                        h_radio = LinearCatFn.forward(ctx_cat, self.reduce_dim.weight, self.reduce_dim.bias, *xs)
                    else:
                        h_radio = xs[0]
                    ps_r, cfg = stack_args(self.attention_net_radio)
                    ctx_r = HandCtx((ctx_cat is not None,) + (True,) * 8 + (False,) * 4)
                    _, A_raw["radiology"] = AmilPoolFn.forward(ctx_r, h_radio, *ps_r, *cfg, M_out=slot("radio"))
            if "path" in order:
                ps_p, cfg = stack_args(self.attention_net_WSI)
                ctx_p = HandCtx((False,) + (True,) * 8 + (False,) * 4)
                prev = ops.set_concurrent(True) if fork else None      # see forward(): 224-CU tile plan beside the branches
                try:
                    _, A_raw["pathology"] = AmilPoolFn.forward(ctx_p, path_x, *ps_p, *cfg, M_out=slot("path"))
                finally:
                    if fork:
                        ops.set_concurrent(prev)
            if "omic" in order:
                with branch():
                    X = kwargs["genomic_features"]
                    if X.dim() == 1:
                        X = X.unsqueeze(0)
                    seed = ops.next_dropout_seed() if tr else 0
                    word = ops._seed_word
                    acts = [ops._f32c(X)]
                    nblk = len(self.fc_omic)
                    for i, blk in enumerate(self.fc_omic):
                        lin, adrop = blk[0], blk[2]
                        acts.append(_dense_fwd_raw(acts[-1], lin.weight, lin.bias, "selu", "alpha" if tr else "none",
                                                   adrop.p if tr else 0.0, seed, i, word,
                                                   out=slot("omic") if i == nblk - 1 else None))
            if fork:
                cur.wait_stream(side)
            if self.fusion == "concat":
                # ---- classifier + hazards + loss + their backward: one launch on the branches' slots
                Wk, bk = self.classifier.weight, self.classifier.bias
                dWk, dbk = torch.empty_like(Wk), torch.empty_like(bk)
                hazards, S, Y_hat, loss, risk, dfeat = ops.surv_head_nll_step(feat, Wk, bk, label, c, alpha, dWk, dbk,
                                                                              loss_scale=loss_scale)
                grads[Wk], grads[bk] = dWk, dbk
                dslot = lambda k: dfeat[:, off[k]:off[k] + width[k]]
            else:
                # ---- XlinearFusion (one node's forward / backward bodies, run by hand), classifier[0] + ReLU + Dropout,
                # then classifier[3] + hazards + loss + their backward in one launch (forward() lines 182-188)
                from ..ops import XFusionFn
                seed_f = ops.next_dropout_seed() if tr else 0
                word_f = ops._seed_word
                fus = self.mm
                weights = []
                for i in range(len(order)):
                    for lin in (fus.reduce[i][0][0], fus.reduce[i][1][0], fus.reduce[i][2][0]):
                        weights += [lin.weight, lin.bias]
                weights += [fus.encoder1[0].weight, fus.encoder1[0].bias, fus.encoder2[0].weight, fus.encoder2[0].bias]
                p_f = fus.dropout_rate if tr else 0.0
                ctx_x = HandCtx((False,) * 3 + (True,) * (len(order) + len(weights)))
                MMv = XFusionFn.forward(ctx_x, len(order), p_f, seed_f, *[slot(k) for k in order], *weights)
                c0, c3 = self.classifier[0], self.classifier[3]
                p_c = self.classifier[2].p if tr else 0.0
                kind_c = "dropout" if tr else "none"
                hid = _dense_fwd_raw(MMv, c0.weight, c0.bias, "relu", kind_c, p_c, seed_f & 0xFFFFFFFF, 11, word_f)
                dWk, dbk = torch.empty_like(c3.weight), torch.empty_like(c3.bias)
                hazards, S, Y_hat, loss, risk, dhid = ops.surv_head_nll_step(hid, c3.weight, c3.bias, label, c, alpha,
                                                                             dWk, dbk, loss_scale=loss_scale)
                grads[c3.weight], grads[c3.bias] = dWk, dbk
                dMM, grads[c0.weight], grads[c0.bias] = _dense_bwd_raw(dhid, hid, MMv, c0.weight, True, "relu", kind_c, p_c,
                                                                       seed_f & 0xFFFFFFFF, 11, word=word_f)
                outx = XFusionFn.backward(ctx_x, dMM)
                dvs = dict(zip(order, outx[3:3 + len(order)]))
                for p_, g_ in zip(weights, outx[3 + len(order):]):
                    grads[p_] = g_
                dslot = lambda k: dvs[k]
            if fork:
                side.wait_stream(cur)
            # ---- backward: the pathology stack on this stream, the small branches beside it
            if "path" in order:
                stack_backward(ctx_p, ps_p, dslot("path"))
            if "omic" in order:
                with branch():
                    g = dslot("omic")
                    for i in range(nblk - 1, -1, -1):
                        lin, adrop = self.fc_omic[i][0], self.fc_omic[i][2]
                        g, dW, db = _dense_bwd_raw(g, acts[i + 1], acts[i], lin.weight, lin.bias is not None, "selu",
                                                   "alpha" if tr else "none", adrop.p if tr else 0.0, seed, i,
                                                   need_dx=i > 0, word=word)
                        grads[lin.weight] = dW
                        if lin.bias is not None:
                            grads[lin.bias] = db
            if "radio" in order:
                with branch():
                    dh = stack_backward(ctx_r, ps_r, dslot("radio"))
                    if ctx_cat is not None:
                        out = LinearCatFn.backward(ctx_cat, dh)
                        grads[self.reduce_dim.weight], grads[self.reduce_dim.bias] = out[0], out[1]
            if fork:
                cur.wait_stream(side)
            # ---- hand the gradients over (parameters of branches outside `mode` took no part: no gradient, as in autograd)
            if grad_out is not None:
                dst, src = [], []
                for p, t in zip(params, grad_out):
                    if p in grads:
                        dst.append(t)
                        src.append(grads[p])
                    elif not accumulate:
                        t.zero_()
                if accumulate:
                    torch._foreach_add_(dst, src)
                else:
                    torch._foreach_copy_(dst, src)
            else:
                dst, src = [], []
                for p in params:
                    g = grads.get(p)
                    if g is None:
                        continue
                    if p.grad is None:
                        p.grad = g
                    else:
                        dst.append(p.grad)
                        src.append(g)
                if dst:
                    torch._foreach_add_(dst, src)
        return hazards, S, Y_hat, A_raw, loss, risk

    def forward(self, **kwargs):
        A_raw = {}
        path_x = kwargs.get("path_features") if "path" in self.mode else None
        fork = self._fork_ok(path_x)
        if fork:
            cur = torch.cuda.current_stream(path_x.device)
            side = self._side_stream(path_x.device)
            side.wait_stream(cur)
            branch = lambda: torch.cuda.stream(side)
        else:
            import contextlib
            branch = contextlib.nullcontext
        joined = []
        # python order (and with it the dropout-seed order) stays radio, path, omic, fusion
        if "radio" in self.mode:
            with branch():
                h_radio = [kwargs[m] for m in self.modalities]
                if fork:
                    for t in h_radio:
                        t.record_stream(side)
                if len(self.modalities) > 1:
                    h_radio = ops.linear_cat(h_radio, self.reduce_dim.weight, self.reduce_dim.bias)
                else:
                    h_radio = h_radio[0]
                M_radio, A_raw["radiology"] = amil_stack(self.attention_net_radio, h_radio, self.training)
                joined += [M_radio, A_raw["radiology"]]
        if "path" in self.mode:
            # with the small branches on the side stream the pathology stack plans its wide tiles for 224 CUs (the hint of
            # mmf_amil_desc::concurrent): the branches' kernels then find CUs while a projection / K-dh launch is resident
            # instead of waiting for it to drain (as one hipGraph 1.13 -> 1.05 ms, tools/r4_mm_try.sh)
            prev = ops.set_concurrent(True) if fork else None
            M_path, A_raw["pathology"] = amil_stack(self.attention_net_WSI, kwargs["path_features"], self.training)
            if fork:
                ops.set_concurrent(prev)
        if "omic" in self.mode:
            with branch():
                X = kwargs["genomic_features"]
                if fork:
                    X.record_stream(side)
                if X.dim() == 1:
                    X = X.unsqueeze(0)
                O = snn_stack(self.fc_omic, X, self.training)
                joined.append(O)
        if fork:
            cur.wait_stream(side)
            for t in joined:
                t.record_stream(cur)

        emb = {"radio": M_radio if "radio" in self.mode else None, "path": M_path if "path" in self.mode else None,
               "omic": O if "omic" in self.mode else None}
        v_list = [emb[k] for k in self._concat_order()]

        if self.fusion == "tensor":
            seed = ops.next_dropout_seed() if self.training else 0
            MM = self.mm(v_list=v_list, seed=seed)
            c0, c3 = self.classifier[0], self.classifier[3]
            hid = ops.dense(MM, c0.weight, c0.bias, act="relu", drop_kind="dropout" if self.training else "none",
                            drop_p=self.classifier[2].p if self.training else 0.0, seed=seed, site=11)
            hazards, S, Y_hat = ops.surv_head(hid, c3.weight, c3.bias)
        else:
            MM = torch.cat(v_list, dim=1)
            hazards, S, Y_hat = ops.surv_head(MM, self.classifier.weight, self.classifier.bias)
        if kwargs.get("return_features"):
            return MM
        return hazards, S, Y_hat, A_raw
