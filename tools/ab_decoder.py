"""A/B harness : the round-3 schedule of the mask decoder (133 launches per forward) as functions that can be patched onto
engine.MaskDecoder, to time it against the round-4 schedule on one box.  python tools/ab_decoder.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from typing import Optional
from camouflaged_vlm_amd import hip, spec, synth, host
from camouflaged_vlm_amd import engine
from camouflaged_vlm_amd.engine import *  # noqa
from camouflaged_vlm_amd.engine import Cascade, Precision, implicit_conv_ok
from camouflaged_vlm_amd.hip import ACT_GELU, ACT_NONE, ACT_RELU, H2


class Old:
    def _attn(self, name: str, q: H2, k: H2, v: H2, B: int, nq: int, nk: int, out: torch.Tensor) -> None:
        """transformer_maskdecoder_edge.py:250-272."""
        ws, heads = self.ws, self.g.dec_heads
        I = self.lin[name + ".q_proj"].N
        qp, kp, vp = ws.f32("a_q", B * nq, I), ws.f32("a_k", B * nk, I), ws.f32("a_v", B * nk, I)
        self.gemm(q, self.lin[name + ".q_proj"], B * nq, out_f32=qp)
        self.gemm(k, self.lin[name + ".k_proj"], B * nk, out_f32=kp)
        self.gemm(v, self.lin[name + ".v_proj"], B * nk, out_f32=vp)
        o = ws.f32("a_o", B * nq, I)
        hip.small_attention(qp, kp, vp, o, B, nq, nk, heads, I // heads)
        oh = ws.h2("a_oh", B * nq, I)
        hip.split_f32(o, oh)
        self.gemm(oh, self.lin[name + ".out_proj"], B * nq, out_f32=out)

    def forward(self, feats: torch.Tensor, sparse: torch.Tensor, no_mask: torch.Tensor, gauss: torch.Tensor,
                B: int, taps: Optional[dict] = None) -> torch.Tensor:
        """feats f32 [B*T][C]; sparse f32 [B][2][C] -> low-res mask logits f32 [B][4G][4G] (mask 0, :133-135)."""
        g, ws = self.g, self.ws
        G, C, T = g.grid, g.prompt_embed_dim, g.grid * g.grid
        if self.pe is None:
            self.pe = torch.empty(T, C, device=self.device)
            hip.dense_pe(gauss, G, C, self.pe)
        fh = ws.h2("feats_h", B * T, C)
        hip.split_f32(feats, fh)
        edge_feat = self._upscale(fh, B, G, "embedding_encoder", False, ws.f32("edge_feat", B * 16 * T, C // 8))
        # :150-158 tokens / src
        NT = 6
        queries = ws.f32("queries", B * NT, C)
        queries.view(B, NT, C).copy_(self.tokens)
        keys = ws.f32("keys", B * T, C)
        hip.add_rows(feats, no_mask, 1, B * T, C, out_f32=keys)
        cond_v = ws.h2("cond_v", B * 2, C)
        cond_k = ws.h2("cond_k", B * 2, C)
        hip.split_f32(sparse, cond_v)
        hip.add_rows(sparse, None, 1, B * 2, C, scale=2.0, out_h2=cond_k)            # cond + cond_pe (:98-99)
        qh, kh, vh = ws.h2("d_q", B * NT, C), ws.h2("d_k", B * T, C), ws.h2("d_v", B * T, C)
        tq, tk = ws.h2("d_tq", B * NT, C), ws.h2("d_tk", B * NT, C)
        ao_q, ao_k = ws.f32("ao_q", B * NT, C), ws.f32("ao_k", B * T, C)
        hidh = ws.h2("d_hid", B * NT, g.dec_mlp)
        mo = ws.f32("d_mlp", B * NT, C)
        for i in range(g.dec_depth):
            L = f"transformer.layers.{i}."
            ln = lambda n: self.ln[L + n]
            # self attention (:174-180); layer 0 replaces the queries
            if i == 0:
                hip.split_f32(queries, tq)
                self._attn(L + "self_attn", tq, tq, tq, B, NT, NT, ao_q)
                hip.layernorm(ao_q, *ln("norm1"), 1e-5, B * NT, C, out_f32=queries)
            else:
                hip.add_rows(queries, self.tokens, NT, B * NT, C, out_h2=tq)
                hip.split_f32(queries, tk)
                self._attn(L + "self_attn", tq, tq, tk, B, NT, NT, ao_q)
                hip.layernorm(queries, *ln("norm1"), 1e-5, B * NT, C, add=ao_q, add_rows=B * NT, out_f32=queries)
            # tokens -> image (:183-187)
            hip.add_rows(queries, self.tokens, NT, B * NT, C, out_h2=qh)
            hip.add_rows(keys, self.pe, T, B * T, C, out_h2=kh)
            hip.split_f32(keys, vh)
            self._attn(L + "cross_attn_token_to_image", qh, kh, vh, B, NT, T, ao_q)
            hip.layernorm(queries, *ln("norm2"), 1e-5, B * NT, C, add=ao_q, add_rows=B * NT, out_f32=queries)
            # tokens -> cond (:189-193)
            hip.add_rows(queries, self.tokens, NT, B * NT, C, out_h2=qh)
            self._attn(L + "cross_attn_token_to_cond", qh, cond_k, cond_v, B, NT, 2, ao_q)
            hip.layernorm(queries, *ln("norm2_cond"), 1e-5, B * NT, C, add=ao_q, add_rows=B * NT, out_f32=queries,
                          out_h2=qh)
            # MLP (:196-198)
            self.gemm(qh, self.lin[L + "mlp.lin1"], B * NT, out_h2=hidh, act=ACT_RELU)
            self.gemm(hidh, self.lin[L + "mlp.lin2"], B * NT, out_f32=mo)
            hip.layernorm(queries, *ln("norm3"), 1e-5, B * NT, C, add=mo, add_rows=B * NT, out_f32=queries)
            # image -> cond (:201-205): q = keys + pe, k = 2*cond, v = cond
            self._attn(L + "cross_attn_image_to_cond", kh, cond_k, cond_v, B, T, 2, ao_k)
            hip.layernorm(keys, *ln("norm4_cond"), 1e-5, B * T, C, add=ao_k, add_rows=B * T, out_f32=keys)
            # image -> tokens (:208-212)
            hip.add_rows(queries, self.tokens, NT, B * NT, C, out_h2=qh)
            hip.add_rows(keys, self.pe, T, B * T, C, out_h2=kh)
            hip.split_f32(queries, tq)
            self._attn(L + "cross_attn_image_to_token", kh, qh, tq, B, T, NT, ao_k)
            hip.layernorm(keys, *ln("norm4"), 1e-5, B * T, C, add=ao_k, add_rows=B * T, out_f32=keys)
        # final token -> image attention (:103-107)
        hip.add_rows(queries, self.tokens, NT, B * NT, C, out_h2=qh)
        hip.add_rows(keys, self.pe, T, B * T, C, out_h2=kh)
        hip.split_f32(keys, vh)
        self._attn("transformer.final_attn_token_to_image", qh, kh, vh, B, NT, T, ao_q)
        hs = ws.f32("hs", B * NT, C)
        hip.layernorm(queries, *self.ln["transformer.norm_final_attn"], 1e-5, B * NT, C, add=ao_q, add_rows=B * NT,
                      out_f32=hs)
        # :167-170 upscaling + edge feature head
        HW = 16 * T
        m1 = ws.f32("mf_1", B * HW, C // 4)
        edge_emb = ws.f32("edge_emb", B * HW, C // 8)
        if implicit_conv_ok(C // 8):                                 # both 3x3 convolutions as implicit GEMMs
            up_h = ws.h2("upscaled_h", B * HW, C // 8)
            up = self._upscale(vh, B, G, "output_upscaling", True, ws.f32("upscaled", B * HW, C // 8), out_h2=up_h)
            self.gemm(up_h, self.mf[0], B * HW, out_f32=m1, conv3x3=(4 * G, 4 * G, C // 8))
            m1h = ws.h2("mf_1h", B * HW, C // 4)
            hip.layernorm(m1, *self.ln["embedding_maskfeature.1"], 1e-6, B * HW, C // 4, act=ACT_GELU, out_h2=m1h)
            self.gemm(m1h, self.mf[1], B * HW, residual=edge_feat, out_f32=edge_emb, conv3x3=(4 * G, 4 * G, C // 4))
        else:
            up = self._upscale(vh, B, G, "output_upscaling", True, ws.f32("upscaled", B * HW, C // 8))
            col1 = ws.h2("mf_col1", B * HW, 9 * (C // 8))
            hip.im2col3x3(up, B, 4 * G, 4 * G, C // 8, col1)
            self.gemm(col1, self.mf[0], B * HW, out_f32=m1)
            hip.layernorm(m1, *self.ln["embedding_maskfeature.1"], 1e-6, B * HW, C // 4, act=ACT_GELU, out_f32=m1)
            col2 = ws.h2("mf_col2", B * HW, 9 * (C // 4))
            hip.im2col3x3(m1, B, 4 * G, 4 * G, C // 4, col2)
            self.gemm(col2, self.mf[1], B * HW, residual=edge_feat, out_f32=edge_emb)
        # :172-186 hyper-network rows actually used: mask token 0 (hs row 1) and edge token (hs row 5)
        hyper = ws.f32("hyper", B, 5, C // 8)
        row, rowh = ws.f32("h_row", B, C), ws.h2("h_rowh", B, C)
        t1, t2 = ws.h2("h_t1", B, C), ws.h2("h_t2", B, C)
        for tok_row, mlp, slot in ((1, "output_hypernetworks_mlps.0", 0), (5, "edge_mlp", 4)):
            hip.gather_rows(hs, B, NT, C, None, tok_row, row)
            hip.split_f32(row, rowh)
            self.gemm(rowh, self.lin[mlp + ".layers.0"], B, out_h2=t1, act=ACT_RELU)
            self.gemm(t1, self.lin[mlp + ".layers.1"], B, out_h2=t2, act=ACT_RELU)
            self.gemm(t2, self.lin[mlp + ".layers.2"], B, out_f32=hyper[:, slot], ldo=5 * (C // 8))
        low = ws.f32("low", B, HW)
        hip.mask_head(up, edge_emb, hyper, B, HW, C // 8, low)
        if taps is not None:
            taps.update(hs=hs.clone(), src=keys.clone(), upscaled=up.clone(), edge_emb=edge_emb.clone(),
                        hyper=hyper.clone(), low_res_masks=low.clone())
        return low




g, c = spec.DEMO_SAM, spec.DEMO_CLIP
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in synth.make_full_state_dict(g, c).items()}
cas = Cascade(sd, g, c, dev, Precision.named("exact"))
eot = host.eot_for_classes(host.ovcamo_constants()["names_test"].tolist())[:c.n_cls_test]
cas.clip.set_text_bank(cas.clip.text_features(eot, "test"), torch.from_numpy(host.ovcamo_constants()["bank_test"]).float(), "test")
new_forward = engine.MaskDecoder.forward
for B in (1, 8):
    inp, ci, cm = (torch.from_numpy(t).to(dev) for t in synth.make_inputs(g, c, batch=B))
    res = {}
    for rep in range(3):
        for tag in ("old", "new"):
            if tag == "old":
                engine.MaskDecoder.forward = Old.forward
                engine.MaskDecoder._attn = Old._attn
                cas.decoder.pe = cas.decoder.pe if cas.decoder.pe is not None else None
            else:
                engine.MaskDecoder.forward = new_forward
            for _ in range(2):
                m = cas.cascade(inp, ci, cm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 12 if B == 1 else 4
            for _ in range(n):
                m = cas.cascade(inp, ci, cm)
                torch.cuda.synchronize()
            res.setdefault(tag, []).append(1e3 * (time.perf_counter() - t0) / n)
            res[tag + "_mask"] = m[0].clone()
    print(f"B={B}: cascade (not pipelined, one sync per step) old {min(res['old']):.3f} ms  new {min(res['new']):.3f} ms; "
          f"max |mask old - new| = {float((res['old_mask'] - res['new_mask']).abs().max()):.2e}", flush=True)
