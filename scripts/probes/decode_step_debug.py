#!/usr/bin/env python
"""stage-by-stage check of the persistent decode step (csrc/decode_fused.hip) on ONE layer against torch fp32 arithmetic from the
same bf16 values: raw qkv, attention output, gated activation, residual stream, final norm"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dexbotic_amd import kernels as K

dev = "cuda"
torch.manual_seed(0)
d, Hq, Hkv, D, F, T, max_len = [int(v) for v in os.environ.get("DIMS", "256,2,1,128,512,11,20").split(",")]
nq = (Hq + 2 * Hkv) * D
bf = torch.bfloat16
def r(*s, sc=0.05): return (torch.randn(*s, device=dev) * sc).to(bf)
ln1, ln2, fw = (1 + r(d, sc=0.1)), (1 + r(d, sc=0.1)), (1 + r(d, sc=0.1))
wqkv, bqkv, wo, wgu, wd = r(nq, d), r(nq, sc=0.1), r(d, Hq * D), r(2 * F, d), r(d, F)
kc, vc = r(1, Hkv, max_len, D, sc=0.5), r(1, Hkv, max_len, D, sc=0.5)
x = r(d, sc=0.5)
cos, sin = torch.rand(D // 2, device=dev), torch.rand(D // 2, device=dev)
ptrs = [t.data_ptr() for t in (ln1, wqkv, bqkv, wo, ln2, wgu, wd, kc, vc)]
table = torch.tensor(ptrs, dtype=torch.int64).to(dev)
ws = K.decode_step_workspace(d, Hq, Hkv, D, F, dev)
ws.zero_()
out = torch.empty(d, device=dev, dtype=bf)
kc0, vc0 = kc.clone(), vc.clone()
K.decode_step(table, x, out, fw, cos, sin, ws, 1, d, Hq, Hkv, D, F, T, 0, max_len, 1e-6)
torch.cuda.synchronize()
print("timed out:", K.decode_timed_out())
up = lambda v: (v + 127) // 128 * 128
w16 = ws.view(bf)
o0 = 0; xres = w16[o0:o0 + d]; o0 += up(d); qkv = w16[o0:o0 + nq]; o0 += up(nq); ao = w16[o0:o0 + Hq * D]; o0 += up(Hq * D); act = w16[o0:o0 + F]
f = lambda t: t.float()
rb = lambda t: t.to(bf).float()
def norm(xv, w):
    rstd = torch.rsqrt((f(xv) ** 2).mean() + 1e-6)
    return rb(f(w) * rb(f(xv) * rstd))
def err(a, b): return float((f(a) - f(b)).abs().max() / (f(b).abs().max() + 1e-12))
h1 = norm(x, ln1)
qkv_ref = rb(f(wqkv) @ h1 + f(bqkv))
print("qkv      ", err(qkv, qkv_ref))
cc, ss = rb(cos), rb(sin)
def rope(v):
    x1, x2 = v[:D // 2], v[D // 2:]
    return torch.cat([rb(rb(x1 * cc) + rb(-x2 * ss)), rb(rb(x2 * cc) + rb(x1 * ss))])
ao_ref = torch.empty(Hq * D, device=dev)
for h in range(Hq):
    g = h // (Hq // Hkv)
    q = rope(f(qkv[h * D:(h + 1) * D]))
    kn = rope(f(qkv[(Hq + g) * D:(Hq + g + 1) * D]))
    vn = f(qkv[(Hq + Hkv + g) * D:(Hq + Hkv + g + 1) * D])
    Kk = torch.cat([f(kc0[0, g, :T]), kn[None]]); Vv = torch.cat([f(vc0[0, g, :T]), vn[None]])
    p_ = torch.softmax((Kk @ q) / math.sqrt(D), 0)
    ao_ref[h * D:(h + 1) * D] = p_ @ Vv
    if h % (Hq // Hkv) == 0:
        print("  cache k", err(kc[0, g, T], kn), " v", err(vc[0, g, T], vn))
print("attn out ", err(ao, rb(ao_ref)))
x2 = rb(f(wo) @ f(ao) + f(x))
h2 = norm(x2, ln2)
gu = f(wgu) @ h2
g_, u_ = rb(gu[:F]), rb(gu[F:])
act_ref = rb(rb(g_ / (1 + torch.exp(-g_))) * u_)
print("act      ", err(act, act_ref))
x3 = rb(f(wd) @ f(act) + x2)
print("xres     ", err(xres, x3))
print("out      ", err(out, norm(xres, fw)))
if os.environ.get("VERBOSE"):
    a, b = f(qkv), qkv_ref
    bad = ((a - b).abs() > 0.02 * b.abs().max()).nonzero().flatten().tolist()
    print("qkv rows off:", len(bad), "of", nq, "first", bad[:40])
    print("kernel", a[:12].tolist())
    print("ref   ", b[:12].tolist())
    # is the kernel's value the dot product without bias / of another row?
    nob = f(wqkv) @ h1
    print("no-bias", nob[:12].tolist())
