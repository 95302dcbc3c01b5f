import sys; sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, torch
import acids_transforms_amd as A
from oracle import oracle as O
from conftest import rel_max
dev=torch.device('cuda:0')
g=torch.Generator().manual_seed(0)
x=torch.randn(4,2,20000,generator=g)*0.1
st=A.STFT().to(dev); X=st(x.to(dev))
Xr=O.stft_forward(x,O.hann_window(1024),1024,256)
print("stft fwd", rel_max(X.cpu().numpy(), Xr.numpy()))
print("istft", rel_max(st.invert(X).cpu().numpy(), O.istft(Xr.reshape(8,-1,513),O.hann_window(1024),1024,256).reshape(4,2,-1).numpy()))
d=A.DGT().to(dev); Xd=d(x.to(dev)); Xdr=O.stft_forward(x,O.gauss_window(1024),1024,256)
print("dgt fwd", rel_max(Xd.cpu().numpy(), Xdr.numpy()))
for nm in (128,513):
    mg=A.Magnitude(n_mels=nm,mode="unipolar").to(dev); mg.scale_data(X)
    fwd,inv=O.magnitude_banks(O.melscale_fbanks(513,0.0,22050.0,nm,44100))
    off,sc=O.magnitude_scale_stats(Xr,"log1p","unipolar")
    y=mg(X); yr=O.magnitude_forward(Xr,fwd,"log1p",off,sc)
    print("mel",nm,"banded", rel_max(y.cpu().numpy(), yr.numpy()))
    yf=mg.forward_fused(st,x.to(dev)); print("mel",nm,"fused", rel_max(yf.cpu().numpy(), yr.numpy()))
    yi=mg.invert(y); print("mel",nm,"invert", rel_max(yi.cpu().numpy(), O.magnitude_invert(yr,inv,"log1p",off,sc).numpy()))
f=A.IF(mode=None,method="forward"); print("IF fwd", rel_max(f(X).cpu().numpy(), O.inst_freq(Xr,"forward").numpy()))
print("unwrap", rel_max(A.Phase(unwrap=True)(X).cpu().numpy(), O.unwrap(Xr.angle()).numpy()))
Xc = X.cpu()                     # same spectrum on both sides: isolates the scan kernel from the FFT's rounding
print("IF fwd (same X)", rel_max(f(X).cpu().numpy(), O.inst_freq(Xc, "forward").numpy()))
u = A.Phase(unwrap=True)(X).cpu(); ur = O.unwrap(Xc.angle())
Xr = Xc
dd = (u - ur)
bad_cols = (dd.abs() > 1e-3).any(-2)
k = torch.round(dd / (2 * np.pi))
print("unwrap: columns that differ", int(bad_cols.sum()), "of", bad_cols.numel(), "; all differences are multiples of 2 pi:",
      bool(((dd - 2 * np.pi * k).abs() < 2e-3).all()), "; elements where |jump| is within 1e-5 of pi:",
      int(((Xr.angle()[..., 1:, :] - Xr.angle()[..., :-1, :]).abs() - np.pi).abs().lt(1e-5).sum()))
