"""CAR-HyNet patch descriptor on MI355X -- host side.

Mirrors the reference's interface for this stage (carhynet/models.py):
  * ``CARHyNet`` is an ``nn.Module`` with the reference's 136 state tensors (``CAR_HyNet().state_dict()`` names and shapes,
    models.py:311-362), so ``load_state_dict(torch.load('car_hynet.pth'))`` works unchanged; ``forward(x)`` takes the
    reference's NCHW float input and returns L2-normalised [N, 128] descriptors (eval mode only, models.py:379-399);
  * ``compute_des_batches(patches, color=True)`` takes NHWC patches in [0, 1] like ``HyNetnetFeature2D`` (models.py:655-666)
    and returns a float32 NumPy array.
All arithmetic runs in libgims_hip.so: activations are NHWC f32 in HBM, every 3x3 convolution is the split-bf16x3 GEMM (``gims_linear``:
f32-class accuracy) reading its operand rows straight from the 3x3 neighbourhood of the SPL32 pixel rows (GIMS_LINEAR_CONV3;
only the 3-channel first layer materialises them with ``gims_ch_im2col3``), the 8x8 convolution is that GEMM on the flattened 8x8x128
activation, 1x1 convolutions are ``gims_linear`` in f32, and FRN / TLU / CoordAtt / depthwise stages are the ``gims_ch_*``
kernels.  BatchNorm (eval) is folded into the neighbouring weights.  No CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .synth import carhynet_state_dict_spec

BN_EPS = 1e-5
EPS_L2_NORM = 1e-10


class CARHyNet(nn.Module):
    chunk = 8192                      # patches per pass (32 per CU and launch: the start-time stagger of the per-patch kernels and the launch tails amortise)
    fused_sandglass = True            # False: the layer-by-layer kernels (kept as the cross-check of the fused one)
    fused_frn = True                  # likewise for the FRN (+ CoordAtt) + TLU block
    fused_conv = True                 # 3x3 convolution + FRN block of layers 2-6 as ONE per-patch kernel (gims_ch_conv_block); False: GEMM + FRN block

    def __init__(self):
        super().__init__()
        for name, shape in carhynet_state_dict_spec():
            t = torch.zeros(shape, dtype=torch.int64 if name.endswith("num_batches_tracked") else torch.float32)
            self.register_buffer(name.replace(".", "__"), t)
        self._names = [n for n, _ in carhynet_state_dict_spec()]
        self._pack = None

    # state_dict with the reference's dotted names
    def state_dict(self, *a, **k):
        return {n: getattr(self, n.replace(".", "__")) for n in self._names}

    def load_state_dict(self, sd, strict=True):
        missing = [n for n in self._names if n not in sd]
        extra = [n for n in sd if n not in self._names]
        if strict and (missing or extra):
            raise RuntimeError(f"CARHyNet.load_state_dict: missing {missing[:3]}..., unexpected {extra[:3]}...")
        for n in self._names:
            if n in sd:
                getattr(self, n.replace(".", "__")).copy_(torch.as_tensor(np.asarray(sd[n])) if not torch.is_tensor(sd[n]) else sd[n])
        self._pack = None

    # ------------------------------------------------------------------ weight preparation
    def _prepare(self, dev):
        if self._pack is not None and self._pack["dev"] == dev:
            return self._pack
        sd = {n: getattr(self, n.replace(".", "__")).detach().double().cpu() for n in self._names}
        f32 = lambda t: t.float().contiguous().to(dev)       # noqa: E731

        def bn_fold(p, affine=True):                # y = x * sc + sh
            sc = 1.0 / torch.sqrt(sd[p + "running_var"] + BN_EPS)
            if affine:
                sc = sc * sd[p + "weight"]
            sh = -sd[p + "running_mean"] * sc + (sd[p + "bias"] if affine else 0.0)
            return sc, sh

        def conv3(p, cin_pad=None):                 # [O][I][3][3] -> SPL32 [O][2 * kpad], column (ky*3+kx)*Ipad + i
            w = sd[p + "weight"]
            o, i = w.shape[0], w.shape[1]
            ip = cin_pad or i
            wk = torch.zeros(o, 3, 3, ip, dtype=torch.float64)
            wk[..., :i] = w.permute(0, 2, 3, 1)
            k = 9 * ip
            kpad = (k + 31) // 32 * 32
            full = torch.zeros(o, kpad, dtype=torch.float64)
            full[:, :k] = wk.reshape(o, k)
            wp = hip.pack_conv3_fragments(w).to(dev) if (i % 16 == 0 and o % 32 == 0) else None   # fragment order of gims_ch_conv_block
            wp16 = None
            if i < 16 and o % 32 == 0:                # the first layer: input channels zero-padded to one 16-channel K step
                w16 = torch.zeros(o, 16, 3, 3, dtype=torch.float64)
                w16[:, :i] = w
                wp16 = hip.pack_conv3_fragments(w16).to(dev)
            return dict(w=hip.split_spl32(f32(full)), b=f32(sd[p + "bias"]), kpad=kpad, n=o, wp=wp, wp16=wp16)

        def frn(p, cpad=None):
            c = sd[p + "weight"].numel()
            cp = cpad or c
            w, b = torch.zeros(cp, dtype=torch.float64), torch.zeros(cp, dtype=torch.float64)
            w[:c], b[:c] = sd[p + "weight"].reshape(-1), sd[p + "bias"].reshape(-1)
            return dict(w=f32(w), b=f32(b), eps=float(sd[p + "eps"].abs()))

        def tau(p, cpad=None):
            c = sd[p + "tau"].numel()
            t = torch.zeros(cpad or c, dtype=torch.float64)
            t[:c] = sd[p + "tau"].reshape(-1)
            return f32(t)

        def coordatt(p):
            sc, sh = bn_fold(p + "bn1.")
            w1 = sd[p + "conv1.weight"][:, :, 0, 0] * sc[:, None]
            b1 = sd[p + "conv1.bias"] * sc + sh
            return dict(w1=f32(w1), b1=f32(b1), wh=f32(sd[p + "conv_h.weight"][:, :, 0, 0]), bh=f32(sd[p + "conv_h.bias"]),
                        ww=f32(sd[p + "conv_w.weight"][:, :, 0, 0]), bw=f32(sd[p + "conv_w.bias"]))

        def dw(pw, pbn):                            # depthwise [C][1][3][3] + BN -> wt [9][C], bias [C]
            sc, sh = bn_fold(pbn)
            w = sd[pw + "weight"][:, 0] * sc[:, None, None]
            return dict(wt=f32(w.permute(1, 2, 0).reshape(9, -1)), b=f32(sh))

        def pw(pconv, pbn, o_pad=None, i_pad=None):  # 1x1 conv (no bias) + BN -> f32 [O][I], bias [O]; zero-padded to the GEMM's K % 32
            sc, sh = bn_fold(pbn)
            w = sd[pconv + "weight"][:, :, 0, 0] * sc[:, None]
            o, i = w.shape
            wp, bp = torch.zeros(o_pad or o, i_pad or i, dtype=torch.float64), torch.zeros(o_pad or o, dtype=torch.float64)
            wp[:o, :i], bp[:o] = w, sh
            return dict(w=f32(wp), b=f32(bp))

        def sandglass(p):
            p0, p1 = pw(p + "conv.2.", p + "conv.3."), pw(p + "conv.4.0.", p + "conv.4.1.")
            d0, d1, ca = dw(p + "conv.0.0.", p + "conv.0.1."), dw(p + "conv.5.", p + "conv.6."), coordatt(p + "conv.1.")
            ptrs = [d0["wt"], d0["b"], ca["w1"], ca["b1"], ca["wh"], ca["bh"], ca["ww"], ca["bw"], p0["w"], p0["b"], p1["w"], p1["b"], d1["wt"], d1["b"]]
            return dict(ptrs=ptrs, dw0=d0, ca=ca, mid=dict(w0=p0["w"], b0=p0["b"], w1=p1["w"], b1=p1["b"]), dw1=dw(p + "conv.5.", p + "conv.6."))

        sc7, sh7 = bn_fold("layer7.2.", affine=False)
        w7 = sd["layer7.1.weight"].permute(0, 2, 3, 1).reshape(128, 8 * 8 * 128) * sc7[:, None]      # column (y*8+x)*128 + c: NHWC flatten
        P = dict(dev=dev,
                 l1=dict(frn0=frn("layer1.0.", 4), tau0=tau("layer1.1.", 4), conv=conv3("layer1.2.", 4), frn=frn("layer1.3."), ca=coordatt("layer1.4."),
                         tau=tau("layer1.5.")),
                 l2=dict(conv=conv3("layer2.0."), frn=frn("layer2.1."), ca=coordatt("layer2.2."), tau=tau("layer2.3.")),
                 sg2=sandglass("layer2_5."), sg4=sandglass("layer4_5."),
                 l3=dict(conv=conv3("layer3.0."), frn=frn("layer3.1."), tau=tau("layer3.2.")),
                 l4=dict(conv=conv3("layer4.0."), frn=frn("layer4.1."), tau=tau("layer4.2.")),
                 l5=dict(conv=conv3("layer5.0."), frn=frn("layer5.1."), tau=tau("layer5.2.")),
                 l6=dict(conv=conv3("layer6.0."), frn=frn("layer6.1."), tau=tau("layer6.2.")),
                 l7=dict(w=hip.split_spl32(f32(w7)), b=f32(sh7)))
        self._zeros = torch.zeros(256, dtype=torch.bfloat16, device=dev)      # the out-of-image operand row of GIMS_LINEAR_CONV3
        self._pack = P
        return P

    # ------------------------------------------------------------------ layers
    @staticmethod
    def _spl(rows, c, dev):
        """SPL32 split-bf16 pixel rows of an NHWC activation: [rows, 2c]."""
        return torch.empty((rows, 2 * c), dtype=torch.bfloat16, device=dev)

    @staticmethod
    def _conv3_im2col(x, L, stride):
        """First layer only (4 input channels: not a multiple of 32): materialised 3x3 neighbourhoods + GEMM."""
        n, h, w, c = x.shape
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        cols = torch.empty((n * ho * wo, 2 * L["kpad"]), dtype=torch.bfloat16, device=x.device)
        hip.ch_im2col3(x, stride, cols, L["kpad"])
        out = torch.empty((n * ho * wo, L["n"]), dtype=torch.float32, device=x.device)
        hip.linear(cols, L["w"], spl=True, bias=L["b"], precision=hip.PREC_BF16X3, out=out)
        return out.view(n, ho, wo, L["n"])

    def _conv3(self, xs, n, h, w, L, stride):
        """3x3 convolution straight from the split-bf16 pixel rows xs [n*h*w, 2C] (GIMS_LINEAR_CONV3: no im2col buffer)."""
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        out = torch.empty((n * ho * wo, L["n"]), dtype=torch.float32, device=xs.device)
        args = hip.linear_args(xs, L["w"], a1=self._zeros, bias=L["b"], out=out, precision=hip.PREC_BF16X3, spl=True, conv=(h, w, stride), m=n * ho * wo)
        hip._check(hip.load().gims_linear(hip.C.byref(args), hip._stream()), "gims_linear(conv3)")
        return out.view(n, ho, wo, L["n"])

    @staticmethod
    def _frn_scale(x, F):
        n, h, w, c = x.shape
        return hip.ch_frn_stats(x, F["w"], F["eps"], torch.empty((n, c), dtype=torch.float32, device=x.device))

    @staticmethod
    def _gates(x, s, b, G):
        n, h, w, c = x.shape
        ph = torch.empty((n, h, c), dtype=torch.float32, device=x.device)
        pw = torch.empty((n, w, c), dtype=torch.float32, device=x.device)
        hip.ch_pool_hw(x, s, b, ph, pw)
        ah, aw = torch.empty_like(ph), torch.empty_like(pw)
        hip.ch_gates(ph, pw, G, ah, aw)
        return ah, aw

    def _frn_tlu(self, x, F, tau, G=None, split=False, split_out=None):
        """FRN (+ CoordAtt) + TLU; split=True: the result as SPL32 pixel rows for the next convolution instead of f32."""
        n, h, w, c = x.shape
        if self.fused_frn and (h, w, c) in ((32, 32, 32), (16, 16, 64), (8, 8, 128)):
            # one workgroup per patch, the raw convolution output read once into LDS (gims_ch_frn_block)
            if split:
                return hip.ch_frn_block(x, F, tau, G, None, split_out if split_out is not None else self._spl(n * h * w, c, x.device))
            return hip.ch_frn_block(x, F, tau, G, torch.empty_like(x), None)
        ah = aw = None
        if G is not None:
            # one pass over the raw convolution output serves FRN's statistics and CoordAtt's two pools; the FRN affine map is
            # applied to the pooled values on their way into the gate MLP (the mean of an affine map is the affine map of the mean)
            ph = torch.empty((n, h, c), dtype=torch.float32, device=x.device)
            pw = torch.empty((n, w, c), dtype=torch.float32, device=x.device)
            rowsq = torch.empty_like(ph)
            hip.ch_pool_hw(x, None, None, ph, pw, rowsq)
            s = hip.ch_frn_from_rows(rowsq, w, F["w"], F["eps"], torch.empty((n, c), dtype=torch.float32, device=x.device))
            ah, aw = torch.empty_like(ph), torch.empty_like(pw)
            hip.ch_gates(ph, pw, G, ah, aw, frn_scale=s, frn_bias=F["b"])
        else:
            s = self._frn_scale(x, F)
        if split:
            return hip.ch_apply(x, s, F["b"], ah, aw, tau, None, split_out if split_out is not None else self._spl(n * h * w, c, x.device))
        return hip.ch_apply(x, s, F["b"], ah, aw, tau, torch.empty_like(x))

    def _sandglass_plus(self, x1, S):
        """x1 + SandGlass(x1) = 2 x1 + conv-stack(x1)  (models.py:226-233 adds x1 inside, 383-385 / 387-389 add it again);
        returned as SPL32 pixel rows (it only feeds the next 3x3 convolution)."""
        n, h, w, c = x1.shape
        if self.fused_sandglass and (h, w, c) in ((32, 32, 32), (16, 16, 64)):
            return hip.ch_sandglass(x1, S, self._spl(n * h * w, c, x1.device))       # one workgroup per patch, activation resident in LDS
        y = hip.ch_dwconv3(x1, S["dw0"]["wt"], S["dw0"]["b"], torch.empty_like(x1), relu6_out=True)
        ah, aw = self._gates(y, None, None, S["ca"])
        rows = n * h * w
        z = hip.ch_gate_pw_pw(y, ah, aw, S["mid"], torch.empty_like(y))        # gates applied, 1x1 C->16 (+BN), 1x1 16->C (+BN, ReLU6)
        return hip.ch_dwconv3(z, S["dw1"]["wt"], S["dw1"]["b"], None, res=x1, res_scale=2.0, y_split=self._spl(rows, c, x1.device))

    @torch.no_grad()
    def _features(self, patches, out):
        """patches: [n, 32, 32, 3] f32 on the GPU -> layer 6 output (models.py:380-392) written into `out` [n*64, 256]."""
        if patches.device.type != "cuda":
            raise hip.GimsHipError("CARHyNet runs on the GPU only (no CPU fallback): move the patches to 'cuda'")
        P = self._prepare(patches.device)
        n = patches.shape[0]
        L = P["l1"]
        if self.fused_conv and self.fused_frn and self.fused_sandglass:
            # the whole first layer in one per-patch kernel (gims_ch_conv_block_first)
            xs = hip.ch_conv_block_first(patches.contiguous(), L["frn0"], L["tau0"], L["conv"], L["frn"], L["tau"], L["ca"],
                                         self._spl(n * 1024, 32, patches.device))
            y1 = None
        elif self.fused_frn:
            # FRN(3) + TLU(3) and the first convolution's operand rows in one per-patch kernel (gims_ch_input_block)
            cols = hip.ch_input_block(patches.contiguous(), L["frn0"], L["tau0"], torch.empty((n * 1024, 2 * L["conv"]["kpad"]), dtype=torch.bfloat16, device=patches.device))
            y1 = torch.empty((n * 1024, L["conv"]["n"]), dtype=torch.float32, device=patches.device)
            hip.linear(cols, L["conv"]["w"], spl=True, bias=L["conv"]["b"], precision=hip.PREC_BF16X3, out=y1)
            y1 = y1.view(n, 32, 32, L["conv"]["n"])
        else:
            x = torch.zeros((n, 32, 32, 4), dtype=torch.float32, device=patches.device)      # channel 3 = padding (weights are zero there)
            x[..., :3] = patches
            x = self._frn_tlu(x, L["frn0"], L["tau0"])
            y1 = self._conv3_im2col(x, L["conv"], 1)
        if y1 is not None:
            xs = self._frn_tlu(y1, L["frn"], L["tau"], L["ca"], split=True)
        L = P["l2"]
        dev = patches.device
        if self.fused_conv and self.fused_frn and self.fused_sandglass:
            # layers 2-6: convolution + FRN (+ CoordAtt) + TLU in one per-patch kernel each; the raw convolution outputs never reach HBM
            x1 = hip.ch_conv_block(xs, n, 32, 32, 32, 1, L["conv"], L["frn"], L["tau"], L["ca"], y=torch.empty((n, 32, 32, 32), dtype=torch.float32, device=dev))
            xs = self._sandglass_plus(x1, P["sg2"])
            xs = hip.ch_conv_block(xs, n, 32, 32, 64, 2, P["l3"]["conv"], P["l3"]["frn"], P["l3"]["tau"], y_split=self._spl(n * 256, 64, dev))
            x1 = hip.ch_conv_block(xs, n, 16, 64, 64, 1, P["l4"]["conv"], P["l4"]["frn"], P["l4"]["tau"], y=torch.empty((n, 16, 16, 64), dtype=torch.float32, device=dev))
            xs = self._sandglass_plus(x1, P["sg4"])
            xs = hip.ch_conv_block(xs, n, 16, 64, 128, 2, P["l5"]["conv"], P["l5"]["frn"], P["l5"]["tau"], y_split=self._spl(n * 64, 128, dev))
            return hip.ch_conv_block(xs, n, 8, 128, 128, 1, P["l6"]["conv"], P["l6"]["frn"], P["l6"]["tau"], y_split=out)
        x1 = self._frn_tlu(self._conv3(xs, n, 32, 32, L["conv"], 1), L["frn"], L["tau"], L["ca"])
        xs = self._sandglass_plus(x1, P["sg2"])
        xs = self._frn_tlu(self._conv3(xs, n, 32, 32, P["l3"]["conv"], 2), P["l3"]["frn"], P["l3"]["tau"], split=True)
        x1 = self._frn_tlu(self._conv3(xs, n, 16, 16, P["l4"]["conv"], 1), P["l4"]["frn"], P["l4"]["tau"])
        xs = self._sandglass_plus(x1, P["sg4"])
        xs = self._frn_tlu(self._conv3(xs, n, 16, 16, P["l5"]["conv"], 2), P["l5"]["frn"], P["l5"]["tau"], split=True)
        return self._frn_tlu(self._conv3(xs, n, 8, 8, P["l6"]["conv"], 1), P["l6"]["frn"], P["l6"]["tau"], split=True, split_out=out)

    def _head(self, xs, n):
        """layer7 + desc_l2norm over ALL patches at once.  The [n*64, 2*128] SPL32 pixel rows ARE the SPL32 layout of the
        flattened [n, 8*8*128] activation (32-channel blocks never straddle a pixel): the 8x8 convolution is one GEMM on a
        view (one launch for the whole batch: a chunk alone would fill 16 of the 256 CUs)."""
        P = self._pack
        raw = torch.empty((n, 128), dtype=torch.float32, device=xs.device)
        hip.linear(xs.view(n, 64 * 256), P["l7"]["w"], spl=True, bias=P["l7"]["b"], precision=hip.PREC_BF16X3, out=raw)
        desc = hip.ch_l2norm(raw, EPS_L2_NORM, torch.empty_like(raw))
        return desc, raw

    def _forward_nhwc(self, patches):
        """patches: [N, 32, 32, 3] f32 on the GPU -> (desc [N, 128], raw [N, 128]); the convolution stack runs in chunks."""
        n = patches.shape[0]
        if n == 0:
            z = torch.zeros((0, 128), dtype=torch.float32, device=patches.device)
            return z, z.clone()
        feats = torch.empty((n * 64, 256), dtype=torch.bfloat16, device=patches.device)
        for i in range(0, n, self.chunk):
            m = min(self.chunk, n - i)
            self._features(patches[i:i + m], feats[i * 64:(i + m) * 64])
        return self._head(feats, n)

    def forward(self, x, mode="eval"):
        """x: [N, 3, 32, 32] like the reference's CAR_HyNet.forward (models.py:379); eval mode only."""
        if self.training:
            raise NotImplementedError("CARHyNet: training mode (Dropout, batch statistics) is not on the HIP path")
        with torch.no_grad():
            desc, raw = self._forward_nhwc(x.permute(0, 2, 3, 1).float().contiguous())
        return (desc, raw) if mode == "train" else desc

    def compute_des_batches(self, patches, color=True):
        """HyNetnetFeature2D.compute_des_batches (models.py:655-666): NHWC patches in [0, 1] -> float32 [N, 128] NumPy array."""
        if not color:
            raise NotImplementedError("CARHyNet: the grey-level variant (HyNet, 1 input channel) is not built")
        dev = torch.device("cuda", torch.cuda.current_device())
        with torch.no_grad():
            p = torch.from_numpy(np.ascontiguousarray(patches, dtype=np.float32)).to(dev)
            return self._forward_nhwc(p)[0].cpu().numpy()

    def compute_sift(self, patches, kps, color=True):
        """HyNetnetFeature2D.compute_sift (models.py:668-671): what utils.common.sift_forward calls on ``data['carhynet']``
        (common.py:886) -- so an instance of this class can be handed to the reference's front end as its ``carhynet``."""
        if len(kps) == 0:
            return kps, []
        return kps, self.compute_des_batches(patches, color).astype(np.float32)
