#!/usr/bin/env python3
"""Where does the error of the other input branches (distance features) come from?  CPU-only analysis: the
factorised algebra of oracle/kernel_model_gen.py evaluated in numpy float32 -- as the kernels do, in another summation order -- and the
same with the distance terms taken from centred differences |x_i - x_j|^2, both against float64, next to the reference twin's own
float32-vs-float64 distance.  Test infrastructure (it runs the oracle); output: profiles/r06/gen_conditioning.txt."""
import numpy as np, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import kernel_model_gen as kg, kernel_model as km, synth, reference_twin as twin
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', 'kernel_model_gen.py')).read()
def make(dtype_name, direct):
    s = src.replace('np.float64', dtype_name).replace('from .kernel_model import', 'from oracle.kernel_model import')
    if direct:
        # logits' distance term and D from centred differences (forward only matters for the comparison of energies/forces? use energy + forces via fd not available: compare forward energy)
        s = s.replace('logits = SCALE * (np.einsum("bhid,bhjd->bhij", q, k) + np.einsum("bhic,bjc->bhij", Qx, Kx))',
                      'd2 = ((x[:, :, None, :] - x[:, None, :, :]) ** 2).sum(-1)\n        logits = SCALE * (np.einsum("bhid,bhjd->bhij", q, k) + np.einsum("bhic,bjc->bhij", u, x) + s[..., None] * d2[:, None])')
        s = s.replace('D = x2[:, None] - 2.0 * np.einsum("bic,bhic->bhi", x, m1) + m2', 'D = np.einsum("bhij,bij->bhi", pr, d2)')
    ns = {}
    exec(compile(s, 'gen32', 'exec'), ns)
    return ns
m64 = make('np.float64', False); m32 = make('np.float32', False); m32d = make('np.float32', True)
# layer_norm / gate etc in kernel_model use float64 internally? they operate on given dtype arrays
def rel(a,b): return np.linalg.norm(a.astype(np.float64)-b)/np.linalg.norm(b)
for (N,H,L,xs) in [(10,64,2,1.0),(10,64,2,3.0),(20,64,2,3.0),(20,128,3,3.0),(20,128,3,8.0)]:
  for flags in [(0,1,1),(1,1,0),(0,1,0)]:
    intr,dist,ab = flags
    p = synth.synth_gnn_params(N,H,L,seed=11,decoder_scale=1.0,node_in=N+1+3*ab,edge_in=(3*intr+dist) or 1)
    x = (synth.normal((6,N,3),5,3)*xs); x -= x.mean(1,keepdims=True); t = np.linspace(0.01,0.9,6)
    fw64 = m64['fold_weights'](p,L,N,bool(intr),bool(dist),bool(ab))
    e64,_ = m64['forward'](fw64,x,t)
    fw32 = {k:(v.astype(np.float32) if isinstance(v,np.ndarray) else v) for k,v in fw64.items()}
    fw32['layers']=[{k:(v.astype(np.float32) if isinstance(v,np.ndarray) else v) for k,v in l.items()} for l in fw64['layers']]
    e32,_ = m32['forward'](fw32,x.astype(np.float32),t.astype(np.float32))
    e32d,_ = m32d['forward'](fw32,x.astype(np.float32),t.astype(np.float32))
    fl = tuple(bool(v) for v in flags)
    xt = torch.from_numpy(x.astype(np.float32)); tt = torch.from_numpy(t.astype(np.float32))
    pt = twin.to_torch(p)
    er32 = twin.energy(pt, xt, tt, L, flags=fl).numpy().reshape(e64.shape)
    er64 = twin.energy(twin.to_torch(p,torch.float64), xt.double(), tt.double(), L, flags=fl).numpy().reshape(e64.shape)
    c = e64 - e64.mean()
    sc = np.abs(e64 - e64.mean()).max()
    print(N,H,L,xs,flags, "energy abs err / spread: factorised f32 %.2e  direct f32 %.2e  reference f32 %.2e" % (np.abs(e32-e64).max()/sc, np.abs(e32d-e64).max()/sc, np.abs(er32-er64).max()/sc), "chk %.1e"%(np.abs(er64-e64).max()/sc))
print("---- forces")
for (N,H,L,xs) in [(10,64,2,1.0),(10,64,2,3.0),(20,64,2,3.0),(28,64,2,3.0),(20,128,3,3.0),(20,128,3,8.0)]:
  for flags in [(0,1,1),(1,1,0),(0,1,0),(1,1,1)]:
    intr,dist,ab = flags
    p = synth.synth_gnn_params(N,H,L,seed=11,decoder_scale=1.0,node_in=N+1+3*ab,edge_in=(3*intr+dist) or 1)
    x = (synth.normal((6,N,3),5,3)*xs); x -= x.mean(1,keepdims=True); t = np.linspace(0.01,0.9,6)
    fw64 = m64['fold_weights'](p,L,N,bool(intr),bool(dist),bool(ab))
    e64,st64 = m64['forward'](fw64,x,t); f64 = -m64['backward'](fw64,x,st64)
    fw32 = {k:(v.astype(np.float32) if isinstance(v,np.ndarray) else v) for k,v in fw64.items()}
    fw32['layers']=[{k:(v.astype(np.float32) if isinstance(v,np.ndarray) else v) for k,v in l.items()} for l in fw64['layers']]
    e32,st32 = m32['forward'](fw32,x.astype(np.float32),t.astype(np.float32)); f32 = -m32['backward'](fw32,x.astype(np.float32),st32)
    fl = tuple(bool(v) for v in flags)
    xt = torch.from_numpy(x.astype(np.float32)); tt = torch.from_numpy(t.astype(np.float32))
    r32 = twin.score(twin.to_torch(p), xt, tt, L, flags=fl).numpy(); r64 = twin.score(twin.to_torch(p,torch.float64), xt.double(), tt.double(), L, flags=fl).numpy()
    print(N,H,L,xs,flags,"forces rel err: factorised f32 model %.2e   reference f32 %.2e   ratio %.2f" % (rel(f32,f64), rel(r32,r64), rel(f32,f64)/rel(r32,r64)), "chk %.1e" % rel(r64, f64))
