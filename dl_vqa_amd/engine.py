"""Forward / backward schedule of the VqaNet hot path over the HIP kernels.

This is the host-side mirror of ``models/model.py:53-67`` (VqaNet.forward) and of what autograd
would do for it: a fixed sequence of C-ABI calls (dl_vqa_amd.ops) on torch's current HIP stream.
torch only provides the buffers.  Layouts: images NHWC (channels padded to 4), question
activations time-major [T][B][*], the classifier input `combined` = [weighted v | c_fwd | c_bwd].
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Callable, Dict, Optional

import torch

from . import ops

Tensor = torch.Tensor

# dropout sites (models/model.py:84,156,185,186,194,201,204)
SITE_IMAGE, SITE_TEXT, SITE_ATT_V, SITE_ATT_Q, SITE_ATT_X, SITE_CLS1, SITE_CLS2 = 1, 2, 3, 4, 5, 6, 7
_MASK64 = (1 << 64) - 1


def _site_seed(base: int, site: int) -> int:
    return (base * 0x9E3779B97F4A7C15 + site * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) & _MASK64


class Engine:
    def __init__(self, cfg: dict, embedding_tokens: int, compute_dtype: str = "fp32"):
        t, i, a, c = cfg["text"], cfg["image"], cfg["attention"], cfg["classifier"]
        self.V = embedding_tokens
        self.E = t["embedding_features"]
        self.H = t["question_features"]
        self.ndir = 2 if t["bidirectional"] else 1
        if t["num_lstm_layers"] != 1:
            raise NotImplementedError("num_lstm_layers != 1 (the reference notes it 'needs change of code' too, "
                                      "config.yaml:55)")
        self.channels = list(i["num_channels"])
        self.L = len(self.channels) - 1
        self.stride = i["stride"]
        # kernel_size 3 (config.yaml:58) has the implicit-GEMM / patch kernels; any other size the schema admits
        # (utils/config_schema.py:59) takes the materialised im2col + GEMM form (csrc/conv_generic.hip), fp32 only
        self.ks = int(i["kernel_size"])
        if not 1 <= self.ks <= 15:
            raise ValueError(f"image.kernel_size={self.ks}: 1..15")
        self.mid = a["hidden_dim"]
        self.G = a["glimpses"]
        self.do_option = a["do_option"]
        if self.do_option not in ("+", "*", "|"):
            raise ValueError(f"attention do_option {self.do_option!r} (the reference knows '+', '*', '|')")
        self.att_mode = {"+": 0, "*": 1, "|": 2}[self.do_option]
        self.hid = c["hidden_dim"]
        self.A = cfg["max_answers"]
        self.p_text, self.p_image, self.p_att, self.p_cls = t["dropout"], i["dropout"], a["dropout"], c["dropout"]
        self.C = self.channels[-1]
        self.Q = self.ndir * self.H
        self.GC = self.G * self.C
        self.Dc = self.GC + self.Q
        for name, val in (("embedding_features", self.E), ("question_features", self.H), ("hidden_dim(att)", self.mid),
                          ("hidden_dim(cls)", self.hid)) + tuple((f"num_channels[{k}]", ch) for k, ch in
                                                                 enumerate(self.channels[1:], 1)):
            if val % 4:
                raise ValueError(f"{name}={val}: the HIP kernels need multiples of 4 (16-byte vector loads)")
        if self.channels[0] > 4:
            raise ValueError("input images with more than 4 channels are not supported")
        # bf16 path (BASELINE configs[3]): conv blocks 1.. and the v_conv products on bf16 MFMA (fp32 accumulate),
        # activations between the conv blocks stored as bf16; parameters, LSTM, reductions and the optimiser stay
        # fp32.  Opt-in; fp32 is the parity path.
        # fp32x3: the same fp32 tensors everywhere, but the conv blocks' contractions run on the bf16 matrix cores with every
        # fp32 operand split exactly into three bf16 terms (csrc/x3_core.hpp) -- fp32-level accuracy, opt-in.
        if compute_dtype not in ("fp32", "bf16", "fp32x3"):
            raise ValueError(f"compute_dtype {compute_dtype!r} (fp32, fp32x3 or bf16)")
        if self.ks != 3 and compute_dtype != "fp32":
            raise ValueError(f"compute_dtype {compute_dtype!r} needs image.kernel_size=3 (kernel_size={self.ks} runs in fp32)")
        self.bf16 = compute_dtype == "bf16"
        self.x3 = compute_dtype == "fp32x3"
        self.x3_gemm = self.x3 and os.environ.get("VQA_X3_GEMM", "1") != "0"     # diagnostic switch, read once
        # bf16 path: q_lin / lin1 / lin2 on bf16 MFMA too (BASELINE configs[3] "bf16 conv/FC") when every dimension is a multiple
        # of 8 (16-byte operand rows); VQA_FC16=0 keeps them on the fp32 GEMM
        self.fc16 = (self.bf16 and os.environ.get("VQA_FC16", "1") != "0"
                     and all(d % 8 == 0 for d in (self.Q, self.mid, self.Dc, self.hid, self.A)))
        # bf16 path: the LSTM's non-recurrent products (xg = x . W_ih^T, dW_hh, dW_ih, dx) on bf16 MFMA, operands staged as bf16,
        # fp32 accumulation ("bf16 conv/FC + fp32 LSTM accumulate"); the recurrence stays fp32.  VQA_LSTM16=0: all fp32
        self.lstm16 = self.bf16 and os.environ.get("VQA_LSTM16", "1") != "0" and self.H % 8 == 0
        if self.bf16:
            if self.L < 2 or any(ch % 64 for ch in self.channels[1:]) or self.mid % 8 or self.stride != 1:
                raise ValueError("the bf16 path needs >= 2 conv blocks, stride 1 and channel counts that are multiples "
                                 f"of 64 after the first block (num_channels={self.channels}, stride={self.stride})")

    def _pconv_ok(self, H: int, W: int) -> bool:
        """bf16 path: do the patch convolutions cover every block after the first for an H x W image?"""
        if os.environ.get("VQA_PCONV", "1") == "0" or self.stride != 1:
            return False
        h, w = ops.conv_out_hw(H, W, 1)
        for l in range(1, self.L):
            ci, co = self.channels[l], self.channels[l + 1]
            if not (ops.pconv_supported(h, w, ci, co, 1) and ci % 64 == 0 and ops.pconv_wgrad_supported(h, w, ci, co)):
                return False
            h, w = ops.conv_out_hw(h, w, 1)
        return True

    @staticmethod
    def _rows16(x: Tensor, ld: int, rows: int, cols: int, rows8: int) -> Tensor:
        """bf16 copy [rows8, cols] of the fp32 matrix x (leading dimension ld), rows beyond `rows` zero."""
        if ld != cols or not x.is_contiguous():
            t = torch.empty(rows, cols, dtype=torch.float32, device=x.device)
            ops.add2d(x, ld, None, 0, t, cols, rows, cols)
            x = t
        out = torch.zeros(rows8, cols, dtype=torch.bfloat16, device=x.device) if rows8 != rows else \
            torch.empty(rows, cols, dtype=torch.bfloat16, device=x.device)
        ops.to_bf16(x.view(rows, cols), out=out[:rows])
        return out

    @staticmethod
    def _cols16(x: Tensor, rows: int, cols: int, cols8: int) -> Tensor:
        """bf16 copy [rows, cols8] of the contiguous fp32 matrix x [rows, cols], columns beyond `cols` zero."""
        x = x.reshape(rows, cols)
        if cols8 == cols:
            return ops.to_bf16(x)
        t = torch.zeros(rows, cols8, dtype=torch.float32, device=x.device)
        ops.add2d(x, cols, None, 0, t, cols8, rows, cols)
        return ops.to_bf16(t)

    def _x3_layer(self, x_shape, Co) -> bool:
        """fp32x3 mode: does this conv block (NHWC input shape, output channels) run on the split kernels?"""
        return self.x3 and len(x_shape) == 4 and ops.conv_x3_supported(x_shape[1], x_shape[2], x_shape[3], Co, self.stride)

    def _x3_gemm(self, rows) -> bool:
        """fp32x3 mode: the three v_conv products (rows = B * positions) run on the split GEMM when they are large
        enough to fill the chip with 192 x 128 tiles."""
        return self.x3_gemm and rows >= 192 * 64 and self.mid >= 128 and self.C >= 64

    def _out_shape(self, x, l, fast0):
        """NHWC shape of block l's pooled output for input activation x (the NCHW image for the dedicated first block)."""
        B, H, W = (x.shape[0], x.shape[2], x.shape[3]) if (l == 0 and fast0) else ops.nhwc_shape(x)[:3]
        Hp, Wp = ops.conv_out_hw(H, W, self.stride)
        return (B, Hp, Wp, self.channels[l + 1])

    def _side_streams(self, dev):
        # VQA_STREAMS: 0 = one stream; 1 = the question branch on a side stream, joined before the image branch;
        # 2 (default) = the question branch runs UNDER the convolutions (forward: joined before the attention stage,
        # backward: joined at the end).  VQA_STREAMS_BWD (default: VQA_STREAMS) picks the BACKWARD schedule on its
        # own: under data parallelism "1" joins the BPTT chain before the convolution backward starts, so the 'text'
        # bucket (66 % of the gradient bytes) is all-reduced under ALL of the convolution kernels instead of their
        # last 2-3 ms (bench.py --gpus N measures both).  Same box, interleaved, B=256: 27.33 / 26.99 ms per step for 1 / 2 -- the
        # LSTM step launches leave bubbles (prologue / cell epilogue / launch seams of a 44 us kernel) that
        # convolution workgroups fill.
        mode = os.environ.get("VQA_STREAMS", "2")
        if mode == "0":
            cur = torch.cuda.current_stream(dev)
            return [cur, cur]
        if not hasattr(self, "_sides"):
            self._sides = {}
        key = str(dev)
        if key not in self._sides:
            # (a high-priority side stream measured no difference: 27.03 / 27.05 ms per step, 115.4 / 115.6 at the
            # stress shape)
            self._sides[key] = [torch.cuda.Stream(device=dev) for _ in range(2)]
        return self._sides[key]

    # ------------------------------------------------------------------ stream fork / join
    def _fork_join(self, dev, jobs):
        """Run independent launch sequences concurrently: job 0 on the current stream, the others on side
        streams that first wait for everything enqueued so far; the current stream then waits for them."""
        if len(jobs) == 1:
            jobs[0]()
            return
        main = torch.cuda.current_stream(dev)
        if not hasattr(self, "_side"):
            self._side = {}
        fork = torch.cuda.Event()
        fork.record(main)
        joins = []
        for k, job in enumerate(jobs[1:]):
            side = self._side.setdefault((str(dev), k), torch.cuda.Stream(device=dev))
            side.wait_event(fork)
            with torch.cuda.stream(side):
                job()
                ev = torch.cuda.Event()
                ev.record(side)
                joins.append(ev)
        jobs[0]()
        for ev in joins:
            main.wait_event(ev)

    # ------------------------------------------------------------------ forward
    def forward(self, P: Dict[str, Tensor], v: Tensor, q: Tensor, q_len: Tensor, training: bool, seed: int,
                keep: bool, bad_tokens: Optional[Tensor] = None):
        """Returns (logits [B,A], ctx or None). `keep` = save what backward needs.
        bad_tokens: optional device int32 [1] that counts token ids outside the vocabulary."""
        # kernels launch on HIP's CURRENT device and torch's current stream of that device: make the tensors'
        # device current for the whole schedule (a model on cuda:1 while cuda:0 is current would otherwise
        # launch on GPU 0 with GPU-1 pointers)
        with torch.cuda.device(v.device):
            return self._forward(P, v, q, q_len, training, seed, keep, bad_tokens)

    def _forward(self, P: Dict[str, Tensor], v: Tensor, q: Tensor, q_len: Tensor, training: bool, seed: int,
                 keep: bool, bad_tokens: Optional[Tensor] = None):
        assert v.is_cuda and v.dtype in (torch.float32, torch.float16) and v.dim() == 4, \
            "v must be a float32 (or the dataset's float16) CUDA tensor [B,C,S,S]"
        v = v.contiguous()
        q = q.to(device=v.device, dtype=torch.int64).contiguous()
        q_len = q_len.to(device=v.device, dtype=torch.int64).contiguous()
        dev = v.device
        B, T = q.shape
        E, H, G, C, mid, hid, A, Dc, GC, Q = self.E, self.H, self.G, self.C, self.mid, self.hid, self.A, self.Dc, self.GC, self.Q
        tr = bool(training)
        sd = lambda site: _site_seed(seed, site)
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)

        # ---- question encoder (model.py:155-166)
        p_txt = self.p_text if tr else 0.0
        x_emb = ops.embed_tanh_fwd(q, P["text.embedding.weight"], p_txt, sd(SITE_TEXT), bad_tokens)       # [T,B,E]
        combined = new(B, Dc)
        lstm = [None] * self.ndir
        fused = ops.lstm_step_supported(H) and os.environ.get("VQA_FUSED_LSTM", "1") == "1"
        use_graph = os.environ.get("VQA_GRAPH", "1") == "1"

        def sfx(d):
            return "_reverse" if d else ""

        # bf16 path: the LSTM's non-recurrent products on bf16 MFMA (fp32 accumulation): xg = x . W_ih^T here, dW_hh, dW_ih
        # and dx in backward; x and W_ih staged as bf16 with the embedding width padded to a multiple of 8 (16-byte rows).
        # The recurrence h . W_hh^T, the cells and their gradients stay fp32.  VQA_LSTM16=0: everything fp32.
        l16 = self.lstm16 and (T * B) % 8 == 0
        E8 = (E + 7) // 8 * 8
        x16 = [None]

        def run_question_branch():
            if l16:
                x16[0] = self._cols16(x_emb, T * B, E, E8)
            for d in range(self.ndir):
                xg = new(T * B, 4 * H)
                wih16 = None
                if l16:
                    wih16 = self._cols16(P["text.lstm.weight_ih_l0" + sfx(d)], 4 * H, E, E8)
                    ops.gemm_bf16(x16[0], wih16, xg, T * B, 4 * H, E8, bias1=P["text.lstm.bias_ih_l0" + sfx(d)],
                                  bias2=P["text.lstm.bias_hh_l0" + sfx(d)], tag=10)
                else:
                    ops.gemm(x_emb, P["text.lstm.weight_ih_l0" + sfx(d)], xg, T * B, 4 * H, E,
                             bias1=P["text.lstm.bias_ih_l0" + sfx(d)], bias2=P["text.lstm.bias_hh_l0" + sfx(d)], tag=10)
                lstm[d] = SimpleNamespace(gates=new(T, B, 4 * H), xg=xg, Hs=new(T + 1, B, H), Cs=new(T + 1, B, H), wih16=wih16)
                lstm[d].Hs[T if d else 0].zero_()         # h_0 = c_0 = 0: only the initial slot is read before
                lstm[d].Cs[T if d else 0].zero_()         # it is written
            if fused:
                # the whole recurrence, both directions per launch, as ONE call (a cached hipGraph of T launches)
                dirs = [dict(w_hh=P["text.lstm.weight_hh_l0" + sfx(d)], xg=lstm[d].xg, gates=lstm[d].gates,
                             Hs=lstm[d].Hs, Cs=lstm[d].Cs, c_final=combined[:, GC + d * H:], reverse=bool(d))
                        for d in range(self.ndir)]
                ops.lstm_seq_fwd(dirs, q_len, B, T, H, cf_ld=Dc, use_graph=use_graph)
                return
            for d in range(self.ndir):          # H % 32 != 0: recurrent GEMM + cell kernel per step
                st, w_hh = lstm[d], P["text.lstm.weight_hh_l0" + sfx(d)]
                hg = new(B, 4 * H)
                order = range(T) if d == 0 else range(T - 1, -1, -1)
                for n, t in enumerate(order):
                    si, so = (t, t + 1) if d == 0 else (t + 1, t)
                    cf = combined[:, GC + d * H:] if n == T - 1 else None
                    ops.gemm(st.Hs[si], w_hh, hg, B, 4 * H, H, tag=11)
                    ops.lstm_cell_fwd(st.xg[t * B:(t + 1) * B], hg, st.Cs[si], st.Hs[si], q_len, t, st.gates[t],
                                      st.Cs[so], st.Hs[so], cf, Dc)

        # The question branch is a chain of small (M = B) launches, independent of the image branch until the
        # attention stage; it runs on a side stream, under the convolutions (VQA_STREAMS=2, the default; 1: the main
        # stream waits for it before the convolutions start; 0: everything on one stream).
        main = torch.cuda.current_stream(dev)
        sides = self._side_streams(dev)
        fork = torch.cuda.Event()
        fork.record(main)
        sides[0].wait_event(fork)
        with torch.cuda.stream(sides[0]):
            run_question_branch()
            text_done = torch.cuda.Event()
            text_done.record(sides[0])
        if os.environ.get("VQA_STREAMS", "2") == "1":
            main.wait_event(text_done)
        # ---- image encoder: conv+relu+pool x L (models/model.py:79-84)
        # the first block has a dedicated kernel that reads the NCHW image as is (K = 27 is too thin for the
        # generic implicit GEMM); otherwise the image is converted to NHWC4 once
        fast0 = self.ks == 3 and ops.conv0_supported(v.shape[1], v.shape[2], v.shape[3], self.channels[1], self.stride)
        if v.dtype == torch.float16 and not fast0:
            # the dedicated first-block kernels read the dataset's fp16 features as they are (widened where the LDS patch is
            # staged); only shapes they do not cover take a widened copy through the generic NHWC path
            v = ops.half_to_float(v)
        acts = [v if fast0 else ops.nchw_to_nhwc4(v)]
        idxs, wds = [], []
        # bf16 path: blocks 1.. on the patch convolutions (csrc/conv_patch_bf16.hip: LDS-resident input patch, activations
        # between the blocks channel-blocked "C16") when every block's shape has them; VQA_PCONV=0 keeps the implicit-GEMM
        # kernels (A/B runs, parity tests of both)
        use_pc = self.bf16 and fast0 and self._pconv_ok(v.shape[2], v.shape[3])
        for l in range(self.L):
            w = P[f"image.conv{l}.weight"]
            assert w.shape[0] == self.channels[l + 1]
            # fp32x3: a block whose successor runs on the split kernels writes its output x3-packed (split once per
            # tensor by the producer's epilogue; forward and wgrad of the successor read that form)
            if self.ks != 3:
                wk = ops.convk_pack_weights(w, acts[-1].shape[3])
                pooled, am = ops.convk_fwd(acts[-1], wk, P[f"image.conv{l}.bias"], self.ks, self.stride, tag=l)
                acts.append(pooled)
                idxs.append(am)
                wds.append(wk)
                continue
            nxt_x3 = l + 1 < self.L and self._x3_layer(self._out_shape(acts[-1], l, fast0), self.channels[l + 2])
            if l == 0 and fast0:
                pooled, am = ops.conv0_fwd(v, w, P["image.conv0.bias"],
                                           out_dtype=torch.bfloat16 if self.bf16 else torch.float32, bf16_mfma=self.bf16,
                                           out_packed=nxt_x3, out_c16=use_pc)
                acts.append(pooled)
                idxs.append(am)
                wds.append(None)
                continue
            if self.bf16:
                if l == 0:
                    raise ValueError("the bf16 path needs the dedicated first-block kernel (3-channel NCHW image, "
                                     "W % 4 == 0, 32 or 64 output channels)")
                # bf16 activations in, bf16 out (fp32 out of the last block: the L2 normalisation consumes it)
                if use_pc:
                    wf_img, wd_img = ops.pconv_pack_weights(w, need_wd=keep)
                    pooled, am = ops.pconv_fwd(acts[-1], wf_img, P[f"image.conv{l}.bias"], w.shape[0],
                                               out_dtype=torch.float32 if l == self.L - 1 else torch.bfloat16, tag=l)
                    acts.append(pooled)
                    idxs.append(am)
                    wds.append(wd_img)
                    continue
                wfT, wdT = ops.conv_pack_weights_bf16(w, acts[-1].shape[3], need_wd=keep)
                pooled, am = ops.conv_fwd_bf16(acts[-1], wfT, P[f"image.conv{l}.bias"], self.stride,
                                               out_dtype=torch.float32 if l == self.L - 1 else torch.bfloat16, tag=l)
                acts.append(pooled)
                idxs.append(am)
                wds.append(wdT)
                continue
            x_shape = ops.nhwc_shape(acts[-1])
            x3 = self._x3_layer(x_shape, w.shape[0])
            # fp32 backward-data of blocks 1.. on the patch kernel (csrc/conv_patch_f32.hip: the pre-pool gradient built in LDS once
            # per K-slice instead of routed once per tap by the loaders); VQA_PDGRAD=0 keeps the implicit-GEMM kernel
            # (measured at B = 256, 224 x 224: 64-channel input 3.96 -> 3.69 ms; 128-channel input 3.20 -> 3.32 ms, so blocks whose
            # input has a multiple of 128 channels stay on the implicit-GEMM kernel unless VQA_PDGRAD=2 forces the patch kernel)
            pdg_mode = os.environ.get("VQA_PDGRAD", "1")
            pdg = (keep and l > 0 and not x3 and self.stride == 1 and acts[-1].dim() == 4 and pdg_mode != "0"
                   and (x_shape[3] % 128 != 0 or pdg_mode == "2")
                   and ops.pconvf_supported(x_shape[1], x_shape[2], x_shape[3], w.shape[0]))
            wf, wd = ops.conv_pack_weights(w, x_shape[3], need_wd=(keep and l > 0 and not pdg))
            if pdg:
                wd = ("pconvf", ops.pconvf_pack_weights(w))
            if x3:
                # operands are split once per tensor, not by every workgroup in every K-step: the weights here, the
                # input activation by its producer (or here, when the producer could not: it replaces the fp32 tensor)
                wf, wd = ops.x3_split(wf), (ops.x3_split(wd) if wd is not None else None)
                if acts[-1].dim() == 4:
                    acts[-1] = ops.x3_pack(acts[-1])
            pooled, am = ops.conv_fwd(acts[-1], wf, P[f"image.conv{l}.bias"], self.stride, tag=l, x3=x3,
                                      out_packed=x3 and nxt_x3)
            acts.append(pooled)
            idxs.append(am)
            wds.append(wd)
        pooled = acts[-1]
        Pn = pooled.shape[1] * pooled.shape[2]
        # ---- image dropout + L2 normalisation over channels (model.py:84,56)
        p_img = self.p_image if tr else 0.0
        # attention.drop(v) (model.py:185) is written by the same pass as a second output: the v_conv operand, as
        # bf16 on the bf16 path
        p_att = self.p_att if tr else 0.0
        v_in = v16 = None
        if self.bf16:
            vn, norm, v16 = ops.l2norm_fwd(pooled, p_img, sd(SITE_IMAGE), drop2=(p_att, sd(SITE_ATT_V), torch.bfloat16))
        elif p_att > 0:
            vn, norm, v_in = ops.l2norm_fwd(pooled, p_img, sd(SITE_IMAGE), drop2=(p_att, sd(SITE_ATT_V), torch.float32))
        else:
            vn, norm = ops.l2norm_fwd(pooled, p_img, sd(SITE_IMAGE))
            v_in = vn
        main.wait_event(text_done)

        qf = combined[:, GC:]

        # ---- attention (model.py:183-195): v' = v_conv(drop(v)), q' = q_lin(drop(q)), x = relu(v' + q')
        wv16 = ops.to_bf16(P["attention.v_conv.weight"].view(mid, C)) if self.bf16 else None
        if p_att > 0:
            q_in = new(B, Q)
            ops.add2d(qf, Dc, None, 0, q_in, Q, B, Q)
            ops.dropout(q_in, p_att, sd(SITE_ATT_Q), out=q_in)
            ld_q = Q
        else:
            q_in, ld_q = qf, Dc
        qp = new(B, mid)
        fc = None
        if self.fc16:
            # bf16 FC: activations and weights rounded to bf16 where the GEMM stages them (rows padded to a multiple of 8 with
            # zeros: they become the K dimension of the weight-gradient products), fp32 accumulation, fp32 outputs
            B8 = (B + 7) // 8 * 8
            fc = SimpleNamespace(B8=B8,
                                 wq=ops.to_bf16(P["attention.q_lin.weight"]), w1=ops.to_bf16(P["classifier.lin1.weight"]),
                                 w2=ops.to_bf16(P["classifier.lin2.weight"]))
            fc.q16 = self._rows16(q_in, ld_q, B, Q, B8)
            ops.gemm_bf16(fc.q16, fc.wq, qp, B, mid, Q, bias1=P["attention.q_lin.bias"], tag=20)
        else:
            ops.gemm(q_in, P["attention.q_lin.weight"], qp, B, mid, Q, lda=ld_q, bias1=P["attention.q_lin.bias"], tag=20)
        # x = relu(v' + q') | relu(v' * q') | relu(cat[v', q'])  (model.py:188-193); v' itself is only kept for
        # '*' (its backward needs it), as the aux output of the same GEMM
        # (bf16 path: x is stored as bf16 -- it is streamed three more times and becomes, overwritten in place by
        # d loss / d v', the bf16 operand of both v_conv gradient products)
        xs = torch.empty(B * Pn, mid, dtype=torch.bfloat16 if self.bf16 else torch.float32, device=dev)
        mode = self.att_mode
        vprime = new(B * Pn, mid) if (mode == 1 and keep) else None
        if self.bf16 and vprime is None and os.environ.get("VQA_TALL_GEMM", "1") != "0" and \
                ops.gemm_tall_bf16_supported(B * Pn, mid, C, Pn, mode != 2):
            # tall GEMM with K = C only: persistent 256 x 128 tiles (csrc/gemm_tall_bf16.hip)
            ops.gemm_tall_bf16(v16.view(B * Pn, C), wv16, xs, B * Pn, mid, C, rowgroup=(qp if mode != 2 else None),
                               rg_div=Pn, rg_op=(1 if mode == 1 else 0), relu=True, tag=21)
        elif self.bf16:
            ops.gemm_bf16(v16.view(B * Pn, C), wv16, xs, B * Pn, mid, C, rowgroup=(qp if mode != 2 else None),
                          rg_div=Pn, rg_op=(1 if mode == 1 else 0), relu=True, aux=vprime, tag=21)
        else:
            ops.gemm(v_in, P["attention.v_conv.weight"], xs, B * Pn, mid, C, rowgroup=(qp if mode != 2 else None),
                     rg_div=Pn, rg_op=(1 if mode == 1 else 0), relu=True, aux=vprime, tag=21, x3=self._x3_gemm(B * Pn))
        wx = P["attention.x_conv.weight"]
        score = ops.att_score_fwd(xs, wx.view(G, -1), P["attention.x_conv.bias"], B, Pn, p_att, sd(SITE_ATT_X),
                                  qcat=(qp if mode == 2 else None))
        # ---- softmax over positions + weighted sum (model.py:208-221) -> combined[:, :G*C]
        probs = ops.att_apply_fwd(score, vn, combined, Dc)

        # ---- classifier (model.py:198-205)
        p_cls = self.p_cls if tr else 0.0
        c_in = ops.dropout(combined, p_cls, sd(SITE_CLS1)) if p_cls > 0 else combined
        h1 = new(B, hid)
        logits = new(B, A)
        if fc is not None:
            fc.c16 = self._rows16(c_in, Dc, B, Dc, fc.B8)
            ops.gemm_bf16(fc.c16, fc.w1, h1, B, hid, Dc, bias1=P["classifier.lin1.bias"], relu=True, tag=30)
            h1d = ops.dropout(h1, p_cls, sd(SITE_CLS2)) if p_cls > 0 else h1
            fc.h16 = self._rows16(h1d, hid, B, hid, fc.B8)
            ops.gemm_bf16(fc.h16, fc.w2, logits, B, A, hid, bias1=P["classifier.lin2.bias"], tag=31)
        else:
            ops.gemm(c_in, P["classifier.lin1.weight"], h1, B, hid, Dc, bias1=P["classifier.lin1.bias"], relu=True, tag=30)
            h1d = ops.dropout(h1, p_cls, sd(SITE_CLS2)) if p_cls > 0 else h1
            ops.gemm(h1d, P["classifier.lin2.weight"], logits, B, A, hid, bias1=P["classifier.lin2.bias"], tag=31)

        if not keep:
            return logits, None
        ctx = SimpleNamespace(B=B, T=T, Pn=Pn, q=q, q_len=q_len, acts=acts, idxs=idxs, wds=wds, vn=vn, norm=norm,
                              x_emb=x_emb, x16=x16[0], lstm=lstm, v_in=v_in, v16=v16, wv16=wv16, q_in=q_in, ld_q=ld_q, xs=xs, probs=probs,
                              c_in=c_in, h1=h1, h1d=h1d, fast0=fast0, use_pc=use_pc, fc=fc, vprime=vprime, qp=qp, p_img=p_img, p_txt=p_txt, p_att=p_att, p_cls=p_cls,
                              seed=seed, stages=dict(pooled=pooled, score=score, combined=combined))
        return logits, ctx

    # ------------------------------------------------------------------ backward
    def backward(self, P: Dict[str, Tensor], ctx, dlogits: Tensor, Gr: Dict[str, Tensor],
                 on_ready: Optional[Callable[[str], None]] = None) -> None:
        """Writes the gradient of every parameter into Gr[name] (caller-owned, same shapes as P).

        `on_ready(group)` is called after the kernels producing a parameter group have been enqueued
        ('classifier', 'attention', 'text', 'image'): the data-parallel wrapper starts that bucket's
        all-reduce there, overlapping the rest of backward."""
        with torch.cuda.device(dlogits.device):
            self._backward(P, ctx, dlogits, Gr, on_ready)

    def _backward(self, P, ctx, dlogits, Gr, on_ready):
        B, T, Pn = ctx.B, ctx.T, ctx.Pn
        E, H, G, C, mid, hid, A, Dc, GC, Q = self.E, self.H, self.G, self.C, self.mid, self.hid, self.A, self.Dc, self.GC, self.Q
        dev = dlogits.device
        sd = lambda site: _site_seed(ctx.seed, site)
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        ready = on_ready if on_ready is not None else (lambda group: None)

        # dlogits as a GEMM operand needs a leading dimension that is a multiple of 4
        ldA = (A + 3) // 4 * 4
        if ldA != A or not dlogits.is_contiguous() or dlogits.data_ptr() % 16:
            dl = torch.zeros(B, ldA, dtype=torch.float32, device=dev)
            ops.add2d(dlogits, dlogits.stride(0), None, 0, dl, ldA, B, A)
            dlogits = dl

        # ---- classifier
        fc = ctx.fc
        dh1 = new(B, hid)
        dcomb = new(B, Dc)
        if fc is not None:      # bf16 FC: output gradients staged as bf16 (zero rows up to B8: the K of the dW products)
            dl16 = self._rows16(dlogits, ldA, B, A, fc.B8)
            ops.gemm_bf16(dl16, fc.h16, Gr["classifier.lin2.weight"], A, hid, fc.B8, transA=True, transB=False, lda=A,
                          ldb=hid, tag=40)
            ops.colsum(dlogits, B, A, Gr["classifier.lin2.bias"], ld=ldA)
            ops.gemm_bf16(dl16, fc.w2, dh1, B, hid, A, transB=False, lda=A, ldb=hid, tag=41)
            ops.relu_drop_bwd(ctx.h1, dh1, dh1, ctx.p_cls, sd(SITE_CLS2))
            dh16 = self._rows16(dh1, hid, B, hid, fc.B8)
            ops.gemm_bf16(dh16, fc.c16, Gr["classifier.lin1.weight"], hid, Dc, fc.B8, transA=True, transB=False, lda=hid,
                          ldb=Dc, tag=42)
            ops.colsum(dh1, B, hid, Gr["classifier.lin1.bias"])
            ops.gemm_bf16(dh16, fc.w1, dcomb, B, Dc, hid, transB=False, lda=hid, ldb=Dc, tag=43)
        else:
            ops.gemm(dlogits, ctx.h1d, Gr["classifier.lin2.weight"], A, hid, B, transA=True, transB=False, lda=ldA,
                     ldb=hid, tag=40)
            ops.colsum(dlogits, B, A, Gr["classifier.lin2.bias"], ld=ldA)
            ops.gemm(dlogits, P["classifier.lin2.weight"], dh1, B, hid, A, transB=False, lda=ldA, ldb=hid, tag=41)
            ops.relu_drop_bwd(ctx.h1, dh1, dh1, ctx.p_cls, sd(SITE_CLS2))
            ops.gemm(dh1, ctx.c_in, Gr["classifier.lin1.weight"], hid, Dc, B, transA=True, transB=False, lda=hid, ldb=Dc,
                     tag=42)
            ops.colsum(dh1, B, hid, Gr["classifier.lin1.bias"])
            ops.gemm(dh1, P["classifier.lin1.weight"], dcomb, B, Dc, hid, transB=False, lda=hid, ldb=Dc, tag=43)
        if ctx.p_cls > 0:
            ops.dropout(dcomb, ctx.p_cls, sd(SITE_CLS1), out=dcomb)
        ready("classifier")

        # ---- attention apply + scores
        ds_rows = new(B, G, 1)
        # d loss / d vn has two branches -- the weighted sum (probs x dcomb) and attention.drop(v) -> v_conv -- and one reader,
        # the L2-norm backward: it joins them itself (VQA_JOIN_DVN=0: the three-kernel form att_apply_bwd -> dropout_add ->
        # l2norm_bwd through a [B*P][C] fp32 tensor)
        join = os.environ.get("VQA_JOIN_DVN", "1") != "0"
        dscore, dvn = ops.att_apply_bwd(dcomb, Dc, ctx.probs, ctx.vn, rowsum=ds_rows, want_dvn=not join)
        ops.sum_bgp(ds_rows, Gr["attention.x_conv.bias"])       # x_conv bias gradient: sum over samples of the row sums
        wx = P["attention.x_conv.weight"].view(G, -1)
        dwx_part, dq_part, RS = ops.att_score_bwd(dscore, wx, ctx.xs, B, Pn, ctx.p_att, sd(SITE_ATT_X),
                                                  mode=self.att_mode, vprime=ctx.vprime, qp=ctx.qp)
        dxpre = ctx.xs                                          # overwritten in place: now d loss / d v'
        ops.colsum(dwx_part, B * RS, wx.numel(), Gr["attention.x_conv.weight"])
        dqp = new(B, mid)
        ops.sum_parts(dq_part, dqp, B, RS, mid)
        wv = P["attention.v_conv.weight"]
        if self.bf16:
            # both v_conv gradient products on bf16 MFMA: dW = dx'^T . v_in (both operands reduction-major),
            # dv_in = dx' . Wv (Wv [mid][C] as the [K][N] operand); dx' already is bf16 (written in place over x)
            dx16 = dxpre
            ops.gemm_bf16(dx16, ctx.v16.view(B * Pn, C), Gr["attention.v_conv.weight"].view(mid, C), mid, C, B * Pn,
                          transA=True, transB=False, lda=mid, ldb=C, tag=44)
            if join:
                dv_in = new(B * Pn, C)
                ops.gemm_bf16(dx16, ctx.wv16, dv_in, B * Pn, C, mid, transB=False, lda=mid, ldb=C, tag=45)
            elif ctx.p_att > 0:     # dvn += dropout-mask * (dx' . Wv): one pass joins the two branches
                dv_in = new(B * Pn, C)
                ops.gemm_bf16(dx16, ctx.wv16, dv_in, B * Pn, C, mid, transB=False, lda=mid, ldb=C, tag=45)
                ops.dropout_add(dv_in, dvn, ctx.p_att, sd(SITE_ATT_V))
            else:
                ops.gemm_bf16(dx16, ctx.wv16, dvn.view(B * Pn, C), B * Pn, C, mid, transB=False, lda=mid, ldb=C,
                              accumulate=True, tag=45)
        else:
            gx3 = self._x3_gemm(B * Pn)
            ops.gemm(dxpre, ctx.v_in, Gr["attention.v_conv.weight"], mid, C, B * Pn, transA=True, transB=False, lda=mid,
                     ldb=C, tag=44, x3=gx3)
            if join:
                dv_in = new(B * Pn, C)
                ops.gemm(dxpre, wv, dv_in, B * Pn, C, mid, transB=False, lda=mid, ldb=C, tag=45, x3=gx3)
            elif ctx.p_att > 0:     # dvn += dropout-mask * (dx' . Wv): one pass joins the two branches
                dv_in = new(B * Pn, C)
                ops.gemm(dxpre, wv, dv_in, B * Pn, C, mid, transB=False, lda=mid, ldb=C, tag=45, x3=gx3)
                ops.dropout_add(dv_in, dvn, ctx.p_att, sd(SITE_ATT_V))
            else:
                ops.gemm(dxpre, wv, dvn, B * Pn, C, mid, transB=False, lda=mid, ldb=C, accumulate=True, tag=45, x3=gx3)
        ops.colsum(dqp, B, mid, Gr["attention.q_lin.bias"])
        wq = P["attention.q_lin.weight"]
        if fc is not None:
            dqp16 = self._rows16(dqp, mid, B, mid, fc.B8)
            ops.gemm_bf16(dqp16, fc.q16, Gr["attention.q_lin.weight"], mid, Q, fc.B8, transA=True, transB=False, lda=mid,
                          ldb=Q, tag=46)
            dq_in = new(B, Q)
            ops.gemm_bf16(dqp16, fc.wq, dq_in, B, Q, mid, transB=False, lda=mid, ldb=Q, tag=47)
            if ctx.p_att > 0:
                ops.dropout(dq_in, ctx.p_att, sd(SITE_ATT_Q), out=dq_in)
            ops.add2d(dcomb[:, GC:], Dc, dq_in, Q, dcomb[:, GC:], Dc, B, Q)
        else:
            ops.gemm(dqp, ctx.q_in, Gr["attention.q_lin.weight"], mid, Q, B, transA=True, transB=False, lda=mid,
                     ldb=ctx.ld_q, tag=46)
            if ctx.p_att > 0:
                dq_in = new(B, Q)
                ops.gemm(dqp, wq, dq_in, B, Q, mid, transB=False, lda=mid, ldb=Q, tag=47)
                ops.dropout(dq_in, ctx.p_att, sd(SITE_ATT_Q), out=dq_in)
                ops.add2d(dcomb[:, GC:], Dc, dq_in, Q, dcomb[:, GC:], Dc, B, Q)
            else:
                ops.gemm(dqp, wq, dcomb[:, GC:], B, Q, mid, transB=False, lda=mid, ldb=Q, ldc=Dc, accumulate=True, tag=47)
        ready("attention")

        # ---- LSTM (BPTT over the masked steps), embedding
        dx_parts = [new(T * B, E) for _ in range(self.ndir)]
        fused = ops.lstm_step_supported(H) and os.environ.get("VQA_FUSED_LSTM", "1") == "1"
        use_graph = os.environ.get("VQA_GRAPH", "1") == "1"

        def sfx(d):
            return "_reverse" if d else ""

        def bptt_recurrence():
            """dgates [T][B][4H] of every direction."""
            dcs, dhs = [], []
            for d in range(self.ndir):
                dc = new(B, H)
                ops.add2d(dcomb[:, GC + d * H:], Dc, None, 0, dc, H, B, H)
                dcs.append(dc)
                dhs.append(torch.zeros(B, H, dtype=torch.float32, device=dev))
                ctx.lstm[d].dgates = new(T, B, 4 * H)          # kept alive until the streams have joined
            if fused:
                # one call: the first cell backward + T-1 fused (dgates . W_hh -> next cell backward) launches,
                # every direction per launch, replayed as a cached hipGraph
                dirs = [dict(w_hh=P["text.lstm.weight_hh_l0" + sfx(d)], gates=ctx.lstm[d].gates, Hs=ctx.lstm[d].Hs,
                             Cs=ctx.lstm[d].Cs, dgates=ctx.lstm[d].dgates, dh=dhs[d], dc=dcs[d], reverse=bool(d))
                        for d in range(self.ndir)]
                ops.lstm_seq_bwd(dirs, ctx.q_len, B, T, H, use_graph=use_graph)
                return
            for d in range(self.ndir):
                st, w_hh = ctx.lstm[d], P["text.lstm.weight_hh_l0" + sfx(d)]
                order = range(T - 1, -1, -1) if d == 0 else range(T)
                for n, t in enumerate(order):
                    si, so = (t, t + 1) if d == 0 else (t + 1, t)
                    ops.lstm_cell_bwd(st.gates[t], st.Cs[si], st.Cs[so], ctx.q_len, t, dhs[d], dcs[d], st.dgates[t])
                    if n != T - 1:
                        ops.gemm(st.dgates[t], w_hh, dhs[d], B, H, 4 * H, transB=False, lda=4 * H, ldb=H,
                                 accumulate=True, tag=50)

        def weight_grads(d):
            st = ctx.lstm[d]
            dgates = st.dgates
            h_in = st.Hs[0:T] if d == 0 else st.Hs[1:T + 1]
            l16 = ctx.x16 is not None                # the forward's decision (bf16 path, T * B a multiple of 8)
            if l16:
                # one bf16 copy of dgates feeds the three products; dW_ih and dx come out E8 wide (the padded embedding
                # width) and their first E columns are copied to where the fp32 products would have written them
                E8 = ctx.x16.shape[1]
                dg16 = ops.to_bf16(dgates.view(T * B, 4 * H))
                h16 = ops.to_bf16(h_in.reshape(T * B, H))
                ops.gemm_bf16(dg16, h16, Gr["text.lstm.weight_hh_l0" + sfx(d)], 4 * H, H, T * B, transA=True, transB=False,
                              lda=4 * H, ldb=H, tag=51)
                dwp = Gr["text.lstm.weight_ih_l0" + sfx(d)] if E8 == E else new(4 * H, E8)
                ops.gemm_bf16(dg16, ctx.x16, dwp, 4 * H, E8, T * B, transA=True, transB=False, lda=4 * H, ldb=E8, tag=52)
                if E8 != E:
                    ops.add2d(dwp, E8, None, 0, Gr["text.lstm.weight_ih_l0" + sfx(d)], E, 4 * H, E)
            else:
                ops.gemm(dgates, h_in, Gr["text.lstm.weight_hh_l0" + sfx(d)], 4 * H, H, T * B, transA=True, transB=False,
                         lda=4 * H, ldb=H, tag=51)
                ops.gemm(dgates, ctx.x_emb, Gr["text.lstm.weight_ih_l0" + sfx(d)], 4 * H, E, T * B, transA=True,
                         transB=False, lda=4 * H, ldb=E, tag=52)
            ops.colsum(dgates, T * B, 4 * H, Gr["text.lstm.bias_ih_l0" + sfx(d)])
            ops.add2d(Gr["text.lstm.bias_ih_l0" + sfx(d)], 4 * H, None, 0, Gr["text.lstm.bias_hh_l0" + sfx(d)], 4 * H, 1, 4 * H)
            if l16:
                dxp = dx_parts[d] if E8 == E else new(T * B, E8)
                ops.gemm_bf16(dg16, st.wih16, dxp, T * B, E8, 4 * H, transB=False, lda=4 * H, ldb=E8, tag=53)
                if E8 != E:
                    ops.add2d(dxp, E8, None, 0, dx_parts[d], E, T * B, E)
            else:
                ops.gemm(dgates, P["text.lstm.weight_ih_l0" + sfx(d)], dx_parts[d], T * B, E, 4 * H, transB=False,
                         lda=4 * H, ldb=E, tag=53)

        # The question branch's backward runs on a side stream: BPTT, the weight gradients of every direction, the
        # embedding gradient, and from there the 'text' bucket goes to the data-parallel hook, so its all-reduce
        # overlaps the convolution backward.  Default schedule (VQA_STREAMS=2): the branch runs under the convolution
        # backward and is joined at the end; VQA_STREAMS=1: the main stream waits for it before the convolutions.
        main = torch.cuda.current_stream(dev)
        sides = self._side_streams(dev)
        fork = torch.cuda.Event()
        fork.record(main)
        sides[0].wait_event(fork)
        with torch.cuda.stream(sides[0]):
            bptt_recurrence()
            for d in range(self.ndir):
                weight_grads(d)
            dx_emb = dx_parts[0]
            if self.ndir > 1:
                ops.add(dx_parts[0], dx_parts[1], dx_emb)
            demb = Gr["text.embedding.weight"]       # every row is written (deterministic per-row sums)
            ops.embed_tanh_bwd(ctx.q, ctx.x_emb, dx_emb, demb, ctx.p_txt, sd(SITE_TEXT))
            ready("text")
            ev0 = torch.cuda.Event()
            ev0.record(sides[0])
        if os.environ.get("VQA_STREAMS_BWD", os.environ.get("VQA_STREAMS", "2")) == "1":
            main.wait_event(ev0)

        # ---- image: L2-norm (+dropout) backward, then conv blocks from the last to the first
        c16_hw = tuple(ctx.idxs[-1].shape[2:4]) if self.bf16 and ctx.use_pc else None   # channel-blocked for the routed patches
        if join:
            dP = ops.l2norm_bwd_joined(dcomb, Dc, ctx.probs, dv_in, ctx.p_att, sd(SITE_ATT_V), ctx.vn, ctx.norm, ctx.p_img,
                                       sd(SITE_IMAGE), out_dtype=torch.bfloat16 if self.bf16 else torch.float32, c16_hw=c16_hw)
            if c16_hw is None:
                dP = dP.view_as(ctx.acts[-1])
        elif c16_hw is not None:
            dP = ops.l2norm_bwd(dvn, ctx.vn, ctx.norm, ctx.p_img, sd(SITE_IMAGE), c16_hw=c16_hw)
        else:
            dP = ops.l2norm_bwd(dvn, ctx.vn, ctx.norm, ctx.p_img, sd(SITE_IMAGE),
                                out_dtype=torch.bfloat16 if self.bf16 else torch.float32).view_as(ctx.acts[-1])
        for l in range(self.L - 1, -1, -1):
            if self.ks != 3:
                dP = ops.convk_bwd(ctx.acts[l], dP, ctx.idxs[l], ctx.wds[l], Gr[f"image.conv{l}.weight"], Gr[f"image.conv{l}.bias"],
                                   self.ks, self.stride, need_dx=l > 0, tag=l)
                continue
            if l == 0 and ctx.fast0:
                if self.bf16:
                    ops.conv0_wgrad_bf16(ctx.acts[0], dP, ctx.idxs[0], Gr["image.conv0.weight"], Gr["image.conv0.bias"])
                else:
                    ops.conv0_wgrad(ctx.acts[0], dP, ctx.idxs[0], Gr["image.conv0.weight"], Gr["image.conv0.bias"])
                continue
            if self.bf16 and ctx.use_pc:
                # both kernels route the pre-pool gradient themselves (pooled gradient + arg-max bytes, C16); the block below
                # gets its pooled gradient C16 again, the first block's weight gradient NHWC
                Bx, _, Hx, Wx, _ = ctx.acts[l].shape
                ops.pconv_wgrad(ctx.acts[l], dP, ctx.idxs[l], Gr[f"image.conv{l}.weight"], Gr[f"image.conv{l}.bias"], tag=l)
                dP = ops.pconv_dgrad(dP, ctx.idxs[l], ctx.wds[l], (Bx, Hx, Wx, self.channels[l]), out_c16=l > 1, tag=l)
                continue
            if self.bf16:
                ops.conv_wgrad_bf16(ctx.acts[l], dP, ctx.idxs[l], Gr[f"image.conv{l}.weight"], Gr[f"image.conv{l}.bias"],
                                    self.stride, tag=l)
                dP = ops.conv_dgrad_bf16(dP, ctx.idxs[l], ctx.wds[l], ctx.acts[l].shape, self.stride, tag=l)
                continue
            x_shape = ops.nhwc_shape(ctx.acts[l])
            x3 = self._x3_layer(x_shape, dP.shape[3])
            # fp32x3: the pooled gradient is split once (x3-packed) for its two readers, wgrad and dgrad
            # (one pass: it also sums the bias gradient)
            dPp = ops.x3_pack_pooled_grad(dP, ctx.idxs[l], Gr[f"image.conv{l}.bias"]) if x3 else None
            ops.conv_wgrad(ctx.acts[l], dP, ctx.idxs[l], Gr[f"image.conv{l}.weight"],
                           None if x3 else Gr[f"image.conv{l}.bias"], self.stride, tag=l, x3=x3, dpooled_packed=dPp)
            if l > 0 and isinstance(ctx.wds[l], tuple):
                dP = ops.pconvf_dgrad(dP, ctx.idxs[l], ctx.wds[l][1], x_shape, tag=l)
            elif l > 0:
                dP = ops.conv_dgrad(dPp if x3 else dP, ctx.idxs[l], ctx.wds[l], x_shape, self.stride, tag=l, x3=x3)
        ready("image")
        main.wait_event(ev0)
