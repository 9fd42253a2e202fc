import numpy as np, torch, sys
sys.path.insert(0,'.')
import platipy_amd as pa
from platipy_amd import _lib
from bench import synth_pair
ctx=_lib.Context(0, torch.cuda.current_stream().cuda_stream)
n=256; spacing=(1.0,1.0,1.0)
fixed,moving0,_=synth_pair(ctx,(n,n,n),spacing,4321,torch.device("cuda",0))
c=((n-1)/2.0,)*3
ang=0.05
R=np.array([[np.cos(ang),-np.sin(ang),0],[np.sin(ang),np.cos(ang),0],[0,0,1.0]])
t=(6.0,-4.0,3.0)
mis=pa.AffineTransform(R,t,c)
m0=pa.Image(moving0,spacing)
moving=pa.registration.apply_transform(m0,m0,mis,-1000,pa.sitkLinear)
fi=pa.Image(fixed,spacing)
kw=dict(shrink_factors=[8,4],smooth_sigmas=[0,0],sampling_rate=0.75,optimiser="gradient_descent_line_search")
r_img,r_tfm=pa.registration.linear_registration(fi,moving,reg_method="rigid",**kw)
A,o=r_tfm.matrix_offset()
print("A",np.round(A,4)); print("o",np.round(o,3))
Am,om=mis.matrix_offset()
print("mis A",np.round(Am,4),"o",np.round(om,3))
Ai=np.linalg.inv(Am); oi=-Ai@om
print("mis^-1 A",np.round(Ai,4),"o",np.round(oi,3))
mse=lambda a,b: float(((a-b)**2).mean())
print("mse",mse(fixed,moving.tensor),mse(fixed,r_img.tensor),mse(fixed,moving0))
print(type(r_tfm), [type(x).__name__ for x in getattr(r_tfm,'transforms',[])])
