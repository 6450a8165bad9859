"""CPU simulation (numpy) of the split-operand scheme of the f16x2 mode (csrc/common.h hsplit_t): hi = f16(v), lo = f16(v - hi),
product = hi*hi + lo*hi + hi*lo, with and without flushing of f16 subnormals and with / without a power-of-two pre-scale
of the small operand.  Run before the mode was built (round 5): without flushing 1e-7 ... 2e-6 of the max; with flushing
3e-5 ... 2e-4 -- which is why scripts/probes/mfma_denorm.hip was run on the GPU first (the MFMA does not flush)."""
import numpy as np
rng=np.random.default_rng(0)
def split(v, scale=1.0, flush=False):
    v=(v*scale).astype(np.float32)
    hi=v.astype(np.float16)
    lo=(v-hi.astype(np.float32)).astype(np.float16)
    if flush:
        lo=np.where(np.abs(lo.astype(np.float32))<2.0**-14, np.float16(0), lo)
        hi=np.where(np.abs(hi.astype(np.float32))<2.0**-14, np.float16(0), hi)
    return hi.astype(np.float64), lo.astype(np.float64)
K=9*1024; M=256; N=64
z=rng.standard_normal((M,K)); x=np.maximum(0.1*z,z).astype(np.float32)
for wstd in (0.1,0.01):
  w=np.clip(rng.standard_normal((K,N))*wstd,-2*wstd,2*wstd).astype(np.float32)
  ref=x.astype(np.float64)@w.astype(np.float64)
  for ws in (1.0,64.0):
    for flush in (False,True):
      xh,xl=split(x,1.0,flush); wh,wl=split(w,ws,flush)
      got=(xh@wh+xl@wh+xh@wl)/ws
      got4=got+(xl@wl)/ws
      f16=(xh@wh)/ws
      print(f"wstd {wstd} wscale {ws} flush {flush}: 3-term {np.abs(got-ref).max()/np.abs(ref).max():.2e}  4-term {np.abs(got4-ref).max()/np.abs(ref).max():.2e}  f16 {np.abs(f16-ref).max()/np.abs(ref).max():.2e}")
# gradients: tiny dY
g=(rng.standard_normal((M,N))*1e-5).astype(np.float32)
ref=x.T.astype(np.float64)[:512]@g.astype(np.float64)
for gs in (1.0,1024.0,65536.0):
  for flush in (False,True):
    xh,xl=split(x[:, :512],1.0,flush); gh,gl=split(g,gs,flush)
    got=(xh.T@gh+xl.T@gh+xh.T@gl)/gs
    print(f"wgrad gscale {gs} flush {flush}: {np.abs(got-ref).max()/np.abs(ref).max():.2e}")
