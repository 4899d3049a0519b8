"""Numerical emulation (numpy) of a 3-term split of the Dense layers into 16-bit halves -- hi*hi + hi*lo + lo*hi with
float32 accumulation -- against the float32 and float64 oracles on BASELINE configs[0].  Evidence for DESIGN.md section 4.1b
(rel-Linf vs the float32 restatement, gate 1e-4):
   bfloat16 halves (default):  dense-media weights 1 term 1.2e-2, 3 terms 2.7e-5, 4 terms 1.9e-5;  glorot 3 terms 8.5e-6
   IEEE halves (argument fp16): dense-media weights 1 term 2.0e-3, 3 terms 2.5e-6, 4 terms 2.4e-6;  glorot 3 terms 2.5e-6
Run from the repo root: PYTHONPATH=. python tools/emulate_bf16_split.py [fp16]  (imports oracle/: test infrastructure)."""
import numpy as np, time
from oracle import nerftex_oracle as orc
from nerf_tex_amd import synthetic
def bf16(x):  # round-to-nearest-even to bfloat16, returned as float32
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)
def fp16(x):  # round-to-nearest-even to IEEE half (subnormals kept, as v_mfma_f32_32x32x16_f16 does on gfx950), as float32
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)
FMT = bf16
def split(x):
    hi = FMT(x); lo = FMT((x - hi).astype(np.float32)); return hi, lo
def dense3(x, W, b, relu, terms=3):
    xh, xl = split(x); Wh, Wl = split(W)
    y = xh @ Wh
    if terms >= 3: y = y + xh @ Wl + xl @ Wh
    if terms >= 4: y = y + xl @ Wl
    y = (y + b).astype(np.float32)
    return np.maximum(y, 0) if relu else y
def model3(w, spec, pos, dirs, params, terms):
    f32=np.float32
    g,a=spec.n_geo,spec.n_app
    pm=orc.fourier_features(pos,10,f32); dm=orc.fourier_features(dirs,4,f32)
    pm=np.concatenate([pm,orc.fourier_features(params[:,:g],4,f32)],-1); dm=np.concatenate([dm,orc.fourier_features(params[:,g:],4,f32)],-1)
    h=pm; k=0
    for i in range(8):
        h=dense3(h,w[k],w[k+1],True,terms); k+=2
        if i==4: h=np.concatenate([pm,h],-1)
    alpha=(h@w[k]+w[k+1]).astype(f32); k+=2          # heads stay f32 (VALU)
    h=dense3(h,w[k],w[k+1],False,terms); k+=2
    h=np.concatenate([dm,h],-1)
    h=dense3(h,w[k],w[k+1],True,terms); k+=2
    h=dense3(h,w[k],w[k+1],True,terms); k+=2
    color=(h@w[k]+w[k+1]).astype(f32)
    return color, alpha
g=np.load("tests/golden/golden_plumbing.npz")
spec=orc.ModelSpec(n_parameters=(1,6))
import sys
if len(sys.argv) > 1 and sys.argv[1] == "fp16":
    FMT = fp16
for dense in (True, False):
    blob=synthetic.synthetic_weights(orc.layer_table(spec),seed=0,dense_media=dense); w=orc.split_blob(spec,blob)
    H=W=200;S=32; rows=slice(90*W,110*W)
    ro,rd,t,cone=orc.proxy_rays(orc.full_pixels(H,W)[rows],H,W,float(g["focal"]),g["c2w"],g["b_0"],g["b_1"],np.float32)
    hit=np.isfinite(t[:,0]); ro,rd,t,cone=ro[hit],rd[hit],t[hit],cone[hit]; n=ro.shape[0]
    prm=np.repeat(g["parameters"],n,0)
    r64=orc.render_rays(w,spec,ro,rd,t,prm,cone,S,False,(1,1,1.),dtype=np.float64,return_aux=True)
    r32=orc.render_rays(w,spec,ro,rd,t,prm,cone,S,False,(1,1,1.),dtype=np.float32,return_aux=True)
    ref64=np.concatenate([r64["color_pred"],r64["alpha_pred"][:,None]],-1); ref32=np.concatenate([r32["color_pred"],r32["alpha_pred"][:,None]],-1)
    z=r32["z_vals"]; pts=r32["pts"].reshape(-1,3).astype(np.float32)
    dn=(rd/np.linalg.norm(rd,axis=-1,keepdims=True)).astype(np.float32)
    for terms in (1,3,4):
        c,a=model3(w,spec,pts,np.repeat(dn,S,0),np.repeat(prm,S,0).astype(np.float32),terms)
        cm,am,_,_=orc.map_model_output(c.reshape(n,S,3),a.reshape(n,S),z,rd,False,(1,1,1.),dtype=np.float32)
        got=np.concatenate([cm,am[:,None]],-1)
        print("dense",dense,"terms",terms,"vs f32 oracle %.2e  vs f64 %.2e   (f32 floor %.2e)"%(orc.rel_linf(got,ref32),orc.rel_linf(got,ref64),orc.rel_linf(ref32,ref64)))
