"""CPU emulation (torch fp32) of attention's "Vl·P only where it can matter" rule: for thresholds 2^-k, the share of the (wave, key
tile) pairs whose Vl·P MFMAs the rule skips in every layer, and the largest logit change against always running the pass
(profiles/r05_attention_vl_skip_ab.txt).  usage: python tools/vl_skip_emul.py <weight set> <seed>      e.g.  sens 31"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'zenker-audio-detection_amd'))
from oracle import ast_oracle as orc, ast_torch_cpu as tcpu
from zkast import synth
F=torch.nn.functional
torch.set_num_threads(3)
def f16(x): return x.half().float()
wset, seed = sys.argv[1], int(sys.argv[2])
sd = synth.make_ast_weights(seed, wset)
rec = synth.synth_recording(7, 16000 + 3*8000)
x = torch.from_numpy(orc.extract_features(orc.window_audio(rec), -1.1509622, 3.5340312))
m = tcpu.TorchAST(sd)
def forward(tau):
    B=x.shape[0]; stats=[]
    with torch.inference_mode():
        h = F.conv2d(x.unsqueeze(1).transpose(2,3), m.conv_w, m.conv_b, stride=(10,10)).flatten(2).transpose(1,2)
        h = torch.cat([m.cls.expand(B,-1,-1), m.dist.expand(B,-1,-1), h], 1) + m.pos
        for li,L in enumerate(m.layers):
            y = F.layer_norm(h,(768,),L["ln1"][0],L["ln1"][1],orc.LN_EPS)
            qkv = F.linear(y,L["qkv"][0],L["qkv"][1]).view(B,1214,3,12,64).permute(2,0,3,1,4)
            q,k,v = qkv[0]*0.125, qkv[1], qkv[2]
            s = q @ k.transpose(-1,-2); s = s - s.max(-1,keepdim=True).values
            e = torch.exp(s)                                   # (B,12,1214,1214)
            vh = f16(v); vl = f16(v-vh)
            if tau is None:
                a = (e @ v) / e.sum(-1,keepdim=True)
            else:
                S=1214; pad=(-S)%64
                ep = F.pad(e,(0,pad))                          # keys padded to 19*64
                et = ep.view(B,12,S,19,64)
                tmax = et.max(-1).values                       # (B,12,S,19) tile max weight
                csum = torch.cumsum(et.sum(-1), -1)            # sums through tile t
                prev = torch.cat([torch.zeros_like(csum[...,:1]), csum[...,:-1]], -1)   # sum of tiles < t
                row_skip = tmax <= tau * 0.5 * prev            # per row, conservative half-sum
                rp = (-S)%32
                rs = F.pad(row_skip,(0,0,0,rp), value=True).view(B,12,-1,32,19).all(3)   # wave-uniform (B,12,G,19)
                stats.append(float(rs.float().mean()))
                keep = (~rs).repeat_interleave(32, dim=2)[:,:,:S]                 # (B,12,S,19) tile kept for the row
                keepk = keep.repeat_interleave(64, dim=-1)[..., :S].float()      # per key
                a = (e @ vh + (e*keepk) @ vl) / e.sum(-1,keepdim=True)
            h = h + F.linear(a.transpose(1,2).reshape(B,1214,768), L["o"][0], L["o"][1])
            y = F.layer_norm(h,(768,),L["ln2"][0],L["ln2"][1],orc.LN_EPS)
            h = h + F.linear(F.gelu(F.linear(y,L["fc1"][0],L["fc1"][1])),L["fc2"][0],L["fc2"][1])
        seq = F.layer_norm(h[:,:2],(768,),m.lnf[0],m.lnf[1],orc.LN_EPS)
        z = F.layer_norm((seq[:,0]+seq[:,1])/2,(768,),m.lnh[0],m.lnh[1],orc.LN_EPS)
        return F.linear(z,*m.head).numpy(), stats
ref,_ = forward(None)
for tau in (0.0, 2**-8, 2**-6, 2**-5, 2**-4, 2**-3, 1e9):
    out, st = forward(tau)
    print(f"{wset} tau {tau:.4g}: skip rate per layer {np.round(st,2).tolist()} mean {np.mean(st):.2f}; logit err {np.abs(out-ref).max():.2e}", flush=True)
