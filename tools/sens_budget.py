"""Where the logit error of the parity modes comes from on the input-sensitive `sens` set (CPU emulation, torch fp32).

Each variant rounds ONE class of operands the way the device does and leaves everything else in fp32; the reference is the
plain fp32 forward (oracle/ast_torch_cpu.TorchAST arithmetic).  Windows: the worst ones of tests/test_sens_batch_gpu.py.
usage: python tools/sens_budget.py [window indices ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zenker-audio-detection_amd"))
from oracle import ast_oracle as orc  # noqa: E402
from oracle import ast_torch_cpu as tcpu  # noqa: E402
from zkast import synth  # noqa: E402

F = torch.nn.functional
STAT = {}


def f16(x):
    return x.half().float()


def fp8(x):
    return x.clamp(-448, 448).to(torch.float8_e4m3fn).float()


def lin(x, w, b, mode):
    if mode == "f32":
        return F.linear(x, w, b)
    xh, wh = f16(x), f16(w)
    if mode == "x3":
        xl, wl = f16(x - xh), f16(w - wh)
        return F.linear(xh, wh) + (F.linear(xl, wh) + F.linear(xh, wl)) + b
    if mode == "f16":
        return F.linear(xh, wh, b)
    # c8 with row-scaled activations (LayerNorm planes) is an accuracy refinement; plain c8 here
    e = torch.floor(torch.log2(224.0 / w.abs().max()))
    corr = (F.linear(fp8((x - xh) * 2048.0), fp8(w * 2.0 ** e)) + F.linear(fp8(x), fp8((w - wh) * 2.0 ** (e + 11)))) * 2.0 ** -(e + 11)
    return F.linear(xh, wh) + corr + b


def forward(m, x, gemm="f32", qk="f32", p="f32", v="f32", layers=None, other=None):
    """layers: the encoder layers the rounding modes apply to (None = all); the other layers run `other` (a dict of the same
    keywords, default: everything fp32) — per-layer budgets and the mixed c8 / x3 modes."""
    B = x.shape[0]
    mk = lambda g: g if isinstance(g, dict) else {k: g for k in ("patch", "qkv", "o", "fc1", "fc2")}
    sel = (mk(gemm), qk, p, v)
    oth = (mk((other or {}).get("gemm", "f32")), (other or {}).get("qk", "f32"), (other or {}).get("p", "f32"), (other or {}).get("v", "f32"))
    gm = sel[0]
    gemm = gm.get("patch", "f32")
    with torch.inference_mode():
        h = F.conv2d(x.unsqueeze(1).transpose(2, 3), m.conv_w, m.conv_b, stride=(orc.FSTRIDE, orc.TSTRIDE)) if gemm == "f32" else None
        if h is None:      # patch embedding as the same GEMM modes
            cols = F.unfold(x.unsqueeze(1).transpose(2, 3), 16, stride=10).transpose(1, 2)      # (B,1212,256)
            h = lin(cols, m.conv_w.reshape(768, 256), m.conv_b, gemm)
        else:
            h = h.flatten(2).transpose(1, 2)
        h = torch.cat([m.cls.expand(B, -1, -1), m.dist.expand(B, -1, -1), h], dim=1) + m.pos
        for li, L in enumerate(m.layers):
            gm, qk, p, v = sel if (layers is None or li in layers) else oth
            y = F.layer_norm(h, (768,), L["ln1"][0], L["ln1"][1], orc.LN_EPS)
            qkv = lin(y, L["qkv"][0], L["qkv"][1], gm.get("qkv", "f32")).view(B, orc.SEQ, 3, 12, 64).permute(2, 0, 3, 1, 4)
            q, k, vv = qkv[0] * 0.125, qkv[1], qkv[2]
            kt = k.transpose(-1, -2)
            if qk == "f32":
                s = q @ kt
            else:
                qh, kh = f16(q), f16(k)
                if qk == "x3":
                    s = qh @ kh.transpose(-1, -2) + (f16(q - qh) @ kh.transpose(-1, -2) + qh @ f16(k - kh).transpose(-1, -2))
                elif qk == "f16":
                    s = qh @ kh.transpose(-1, -2)
                else:
                    s = qh @ kh.transpose(-1, -2) + (fp8((q - qh) * 2048.0) @ fp8(k).transpose(-1, -2) + fp8(q) @ fp8((k - kh) * 2048.0).transpose(-1, -2)) * 2.0 ** -11
            s = s - s.max(-1, keepdim=True).values
            e = torch.exp(s)
            if v == "f16":
                vv = f16(vv)
            if p == "f32":
                a = (e / e.sum(-1, keepdim=True)) @ vv
            elif p == "f16":      # the device: fp16 weights, row sum over the rounded weights
                eq = f16(e)
                a = (eq @ vv) / eq.sum(-1, keepdim=True)
            elif p.startswith("thr"):      # the device's running reference: tile-0 row max, moved (for all 32 rows of a wave) when
                thr = float(p[3:])         # some row's tile max exceeds it by more than thr (log2 units); weights fp16, rounded sum
                s2 = (s + s.new_zeros(1)) * 1.4426950408889634      # s already had its row max subtracted: any common shift
                S = s2.shape[-1]
                nrow = s2.shape[-2]
                pad = (-nrow) % 32
                sp = F.pad(s2, (0, 0, 0, pad), value=0.0).view(B, 12, -1, 32, S)      # groups of 32 query rows
                mr = sp[..., :64].max(-1).values                                         # (B,12,G,32)
                refs = torch.empty_like(sp)
                STAT["tiles"] = STAT.get("tiles", 0); STAT["fires"] = STAT.get("fires", 0)
                for t0 in range(0, S, 64):
                    tm = sp[..., t0:t0 + 64].max(-1).values - mr
                    if t0:
                        fire = (tm > thr).any(-1, keepdim=True)
                        STAT["tiles"] += fire.numel(); STAT["fires"] += int(fire.sum())
                        mr = torch.where(fire, mr + tm.clamp(min=0.0), mr)
                    refs[..., t0:t0 + 64] = mr.unsqueeze(-1)
                # a weight rounded at reference r and rescaled later in fp32 keeps its relative rounding error
                eq = f16(torch.exp2(sp - refs)) * torch.exp2(refs - refs[..., -1:])
                eq = eq.view(B, 12, -1, S)[:, :, :nrow]
                a = (eq @ vv) / eq.sum(-1, keepdim=True)
            elif p == "f16_fp32sum":
                a = (f16(e) @ vv) / e.sum(-1, keepdim=True)
            elif p == "bf16x2":   # hi + lo split of the weights (two PV passes per V plane)
                eh = e.bfloat16().float(); el = (e - eh).bfloat16().float(); eq = eh + el
                a = (eq @ vv) / eq.sum(-1, keepdim=True)
            elif p == "f16x2":
                eh = f16(e); el = f16(e - eh); eq = eh + el
                a = (eq @ vv) / eq.sum(-1, keepdim=True)
            h = h + lin(a.transpose(1, 2).reshape(B, orc.SEQ, 768), L["o"][0], L["o"][1], gm.get("o", "f32"))
            y = F.layer_norm(h, (768,), L["ln2"][0], L["ln2"][1], orc.LN_EPS)
            h = h + lin(F.gelu(lin(y, L["fc1"][0], L["fc1"][1], gm.get("fc1", "f32"))), L["fc2"][0], L["fc2"][1], gm.get("fc2", "f32"))
        seq = F.layer_norm(h[:, :2], (768,), m.lnf[0], m.lnf[1], orc.LN_EPS)
        z = F.layer_norm((seq[:, 0] + seq[:, 1]) / 2, (768,), m.lnh[0], m.lnh[1], orc.LN_EPS)
        return F.linear(z, *m.head).numpy()


def stats(out, ref):
    err = np.abs(out - ref).max(axis=1)
    return f"rms {np.sqrt((err ** 2).mean()):.2e}  max {err.max():.2e}"


def per_layer():
    """PER_LAYER=n: rms / max logit error over the first n windows of the test recording for each (layer, source) alone, and
    for the mixed modes "first k layers in x3, the others c8"."""
    n = int(os.environ["PER_LAYER"])
    seed = int(os.environ.get("SEED", "31"))
    sd = synth.make_ast_weights(seed, "sens")
    rec = synth.synth_recording(7, 16000 + (n - 1) * 8000)
    st = (-1.1509622, 3.5340312) if seed == 31 else (-6.5, 2.75)
    x = torch.from_numpy(orc.extract_features(orc.window_audio(rec), *st))
    m = tcpu.TorchAST(sd)
    ref = forward(m, x)
    dev_c8 = dict(gemm=dict(patch="x3", qkv="c8", o="c8", fc1="c8", fc2="c8"), qk="c8", p="thr2")
    dev_x3 = dict(gemm="x3", qk="x3", p="thr2")
    print(f"{n} windows; device f16c8: {stats(forward(m, x, **dev_c8), ref)}; device f16x3: {stats(forward(m, x, **dev_x3), ref)}", flush=True)
    what = os.environ.get("WHAT", "layers,mixed").split(",")
    if "layers" in what:
        for li in range(12):
            row = [f"layer {li:2d}"]
            for name, kw in (("gemm c8", dict(gemm=dict(qkv="c8", o="c8", fc1="c8", fc2="c8"))), ("qk c8", dict(qk="c8")), ("P f16", dict(p="thr2"))):
                row.append(f"{name}: {stats(forward(m, x, layers={li}, **kw), ref)}")
            print("   ".join(row), flush=True)
    if "mixed" in what:
        for k in (1, 2, 3, 4, 6, 8):
            out = forward(m, x, layers=set(range(k)), other=dev_c8, **dict(dev_x3, gemm=dict(patch="x3", qkv="x3", o="x3", fc1="x3", fc2="x3")))
            print(f"first {k} layers x3, rest c8: {stats(out, ref)}", flush=True)
        for k in (8, 6, 4):
            out = forward(m, x, layers=set(range(k, 12)), other=dev_c8, **dev_x3)
            print(f"layers {k}..11 x3, rest c8: {stats(out, ref)}", flush=True)
    if "kinds" in what:
        for kinds in (("qkv",), ("o",), ("fc1",), ("fc2",), ("qkv", "fc2"), ("qkv", "o"), ("fc1", "fc2")):
            g = dict(patch="x3", qkv="c8", o="c8", fc1="c8", fc2="c8")
            for kk in kinds: g[kk] = "x3"
            print(f"c8 but {kinds} x3: {stats(forward(m, x, gemm=g, qk='c8', p='thr2'), ref)}", flush=True)
        print(f"c8 with QK x3: {stats(forward(m, x, gemm=dev_c8['gemm'], qk='x3', p='thr2'), ref)}", flush=True)
        print(f"c8 with P hi+lo: {stats(forward(m, x, gemm=dev_c8['gemm'], qk='c8', p='f16x2'), ref)}", flush=True)
        print(f"x3 with P hi+lo: {stats(forward(m, x, gemm='x3', qk='x3', p='f16x2'), ref)}", flush=True)


def main():
    if os.environ.get("PER_LAYER"):
        return per_layer()
    idx = [int(a) for a in sys.argv[1:]] or [20, 53, 85, 55, 3, 100]
    seed = int(os.environ.get("SEED", "31"))
    sd = synth.make_ast_weights(seed, "sens")
    rec = synth.synth_recording(7, 16000 + 159 * 8000)
    wins = [orc.window_audio(rec)[i] for i in idx]
    st = (-1.1509622, 3.5340312) if seed == 31 else (-6.5, 2.75)
    x = torch.from_numpy(orc.extract_features(wins, *st))
    m = tcpu.TorchAST(sd)
    ref = forward(m, x)
    print("windows", idx, " ref logits", np.round(ref, 3).tolist())
    variants = [("GEMM c8 only", dict(gemm="c8")), ("GEMM x3 only", dict(gemm="x3")), ("QK c8 only", dict(qk="c8")),
                ("QK x3 only", dict(qk="x3")), ("P fp16 (rounded sum) only", dict(p="f16")), ("P fp16, fp32 sum", dict(p="f16_fp32sum")),
                ("P f16 hi+lo", dict(p="f16x2")), ("V fp16 only", dict(v="f16")),
                ("device f16c8 (GEMM c8, QK c8, P f16)", dict(gemm="c8", qk="c8", p="f16")),
                ("device f16x3 (GEMM x3, QK x3, P f16)", dict(gemm="x3", qk="x3", p="f16")),
                ("f16c8 with P hi+lo", dict(gemm="c8", qk="c8", p="f16x2")),
                ("f16x3 with P hi+lo", dict(gemm="x3", qk="x3", p="f16x2"))]
    if os.environ.get("KINDS"):
        variants = [(f"GEMM c8 only in {k}", dict(gemm={k: "c8"})) for k in ("patch", "qkv", "o", "fc1", "fc2")] + [
            ("c8 all but o (x3)", dict(gemm=dict(patch="c8", qkv="c8", o="x3", fc1="c8", fc2="c8"))),
            ("c8 all but qkv (x3)", dict(gemm=dict(patch="c8", qkv="x3", o="c8", fc1="c8", fc2="c8"))),
            ("c8 all but qkv,o (x3)", dict(gemm=dict(patch="c8", qkv="x3", o="x3", fc1="c8", fc2="c8"))),
            ("c8 all", dict(gemm="c8"))]
    else:
      variants = [("P fp16, device reference thr 8", dict(p="thr8")),
                ("P fp16, thr 2", dict(p="thr2")), ("P fp16, thr 1", dict(p="thr1")), ("P fp16, thr 0.5", dict(p="thr0.5")),
                ("P fp16, thr 0", dict(p="thr0")),
                ("device f16c8, thr 8", dict(gemm="c8", qk="c8", p="thr8")), ("device f16c8, thr 1", dict(gemm="c8", qk="c8", p="thr1")),
                ("device f16x3, thr 8", dict(gemm="x3", qk="x3", p="thr8")), ("device f16x3, thr 1", dict(gemm="x3", qk="x3", p="thr1"))] + (
                    variants if os.environ.get("ALL") else [])
    for name, kw in variants:
        STAT.clear()
        out = forward(m, x, **kw)
        if STAT.get("tiles"):
            name = f"{name} [fires {STAT['fires'] / STAT['tiles']:.2f}]"
        err = np.abs(out - ref).max(axis=1)
        print(f"{name:50s} max {err.max():.2e}   per window {np.array2string(err, precision=1, floatmode='fixed', formatter={'float_kind': lambda v: f'{v:.1e}'})}", flush=True)


if __name__ == "__main__":
    main()
