#!/usr/bin/env python3
"""CPU emulation: where does a 16-bit tower's error against the fp32 graph come from, and what is the cheapest operand
format that meets north_star's 1e-3 on trained-like weights?  Every variant rounds weights and / or the
stored activations to sums of n values of a 16-bit type (fp32 accumulate, fp32 heads) and is compared with the float64 graph.
Result (profiles/r04_precision_decomposition.txt): weights and activations contribute about equally, keeping either exact or
the residual stream in fp32 buys < 2x, both split in two halves (three MFMAs per product) buys 800x (fp16) / 80x (bf16)
— k_trunk_split_c128.

    python tools/precision_decomposition.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT+"/tests"):
    sys.path.insert(0,p)
import numpy as np, torch, torch.nn.functional as F
import nethelpers as H
from cchess_zero_amd.net import PolicyValueModule
torch.set_num_threads(8)

def split(t, dt, n):
    """t ~ sum of n pieces of dtype dt (fp32 containers)"""
    out=[]; r=t.clone()
    for i in range(n):
        p=r.to(dt).float(); out.append(p); r=r-p
    return out

def fwd(m, x, mode):
    # mode dict: wdt,wn (weight pieces), adt,an (activation pieces), cross: which products, resid32
    wdt,wn,adt,an=mode["wdt"],mode["wn"],mode["adt"],mode["an"]
    resid32=mode.get("resid32",False)
    def ra(t):
        if adt is None: return t
        return sum(split(t,adt,an))
    def conv(c,h):
        w,b=c.folded()
        if wdt is None and adt is None:
            return F.conv2d(h,w,b,padding=w.shape[-1]//2)
        ws = split(w,wdt,wn) if wdt is not None else [w]
        hs = split(h,adt,an) if adt is not None else [h]
        y=None
        for i,wp in enumerate(ws):
            for j,hp in enumerate(hs):
                if i+j>=max(len(ws),len(hs)): continue   # drop lo*lo
                t=F.conv2d(hp,wp,None,padding=w.shape[-1]//2)
                y=t if y is None else y+t
        return y+b.view(1,-1,1,1)
    h=torch.relu(conv(m.conv_in,x)); hs=ra(h)
    for a,b in m.blocks:
        t=ra(torch.relu(conv(a,hs)))
        y=conv(b,t)+(h if resid32 else hs)
        h=torch.relu(y); hs=ra(h)
    hh = h if resid32 else hs
    wp,bp=m.policy_conv.folded(); wv,bv=m.value_conv.folded()
    p=torch.relu(F.conv2d(hh,wp,bp)).permute(0,2,3,1).reshape(h.shape[0],180)
    v=torch.relu(F.conv2d(hh,wv,bv)).permute(0,2,3,1).reshape(h.shape[0],90)
    return m.policy_fc(p), torch.tanh(m.value_fc2(torch.relu(m.value_fc1(v))))

f16,bf=torch.float16,torch.bfloat16
MODES={
 "fp32": dict(wdt=None,wn=1,adt=None,an=1),
 "fp16 (today)": dict(wdt=f16,wn=1,adt=f16,an=1),
 "bf16": dict(wdt=bf,wn=1,adt=bf,an=1),
 "W fp16 only": dict(wdt=f16,wn=1,adt=None,an=1),
 "A fp16 only": dict(wdt=None,wn=1,adt=f16,an=1),
 "fp16 + fp32 resid": dict(wdt=f16,wn=1,adt=f16,an=1,resid32=True),
 "A fp16x2, W fp16 (2 mfma)": dict(wdt=f16,wn=1,adt=f16,an=2),
 "A fp16, W fp16x2 (2 mfma)": dict(wdt=f16,wn=2,adt=f16,an=1),
 "A fp16, W fp16x2 + fp32 resid": dict(wdt=f16,wn=2,adt=f16,an=1,resid32=True),
 "A fp16x2, W fp16x2 (3 mfma)": dict(wdt=f16,wn=2,adt=f16,an=2),
 "A bf16x2, W bf16x2 (3 mfma)": dict(wdt=bf,wn=2,adt=bf,an=2),
 "A bf16x3, W bf16x3 (6 mfma)": dict(wdt=bf,wn=3,adt=bf,an=3),
}
class _N: pass
for blocks in (7,19):
  for wset in ("trained_like","glorot"):
    n=_N(); n.module=PolicyValueModule(blocks,seed=1); n.refresh=lambda:None
    H.WEIGHT_SETS[wset](n)
    x=torch.from_numpy(H.positions(64,2)).permute(0,3,1,2).contiguous()
    with torch.no_grad():
        m64=PolicyValueModule(blocks,seed=1).double(); m64.load_state_dict({k:v.double() for k,v in n.module.state_dict().items()})
        l64,v64=m64(x.double())
        print("== %d blocks %s: max|logit| %.3g"%(blocks,wset,l64.abs().max()))
        for name,mode in MODES.items():
            l,v=fwd(n.module,x,mode)
            print("  %-34s dlogit %.3g  dvalue %.3g"%(name,(l.double()-l64).abs().max(),(v.double()-v64).abs().max()))
