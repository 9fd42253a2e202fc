import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
import platipy_amd as pa
from platipy_amd import _lib
from bench import synth_pair
from oracle import oracle as O
ctx=_lib.Context(0, torch.cuda.current_stream().cuda_stream)
for n,seed in ((128,1234),(128,4321),(96,1234)):
    shape, spacing=(n,n,n),(1.0,1.0,1.0)
    fixed,moving,_=synth_pair(ctx,shape,spacing,seed,torch.device("cuda",0))
    for variant in ("fused","staged"):
        g_img,g_tfm,g_dvf=pa.registration.fast_symmetric_forces_demons_registration(pa.Image(fixed,spacing),pa.Image(moving,spacing),variant=variant)
        if variant=="fused":
            w_img,w_dvf,_=O.fast_symmetric_forces_demons_registration(O.Vol(fixed.cpu().numpy(),spacing),O.Vol(moving.cpu().numpy(),spacing))
        err=np.abs(g_dvf.numpy()-w_dvf.arr)
        print(n,seed,variant,"median",np.median(err),"p99",np.quantile(err,0.99),"rms",np.sqrt((err**2).mean()),"inner max",err[:,6:-6,6:-6,6:-6].max(),"max",err.max(),"dvfmax",np.abs(w_dvf.arr).max(), "img>0.5", (np.abs(g_img.numpy()-w_img.arr)>0.5).mean())
