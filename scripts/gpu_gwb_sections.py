import os, sys, time, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import headline_array
from pta_replicator_amd import red_noise as rn, device as dv, _lib
from pta_replicator_amd.simulate import make_ideal
psrs, noise = headline_array(68, 5000)
def T(label, f, n=5):
    f(); torch.cuda.synchronize()
    ts=[]
    for _ in range(n):
        t0=time.perf_counter(); r=f(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    print(f"{label:28s} {min(ts)*1e3:7.3f} ms"); return r
grid = T("gwb_time_grid", lambda: rn.gwb_time_grid(psrs, 600, 10))
ORF = T("gwb_orf_device", lambda: rn.gwb_orf_device(psrs, False, [np.sqrt(4*np.pi)], 0))
M = T("cholesky_device", lambda: rn.cholesky_device(ORF.clone()))
Nf=grid["Nf"]
def draws():
    np.random.seed(1)
    w = np.empty((68, Nf, 2))
    for ll in range(68):
        w[ll,:,0]=np.random.randn(Nf); w[ll,:,1]=np.random.randn(Nf)
    return w
w = T("draws", draws)
C = T("gwb_spectrum", lambda: rn.gwb_spectrum(grid["f"], grid["dur"], 10, -15., 13/3., False, 1e-9, 1, 1, None))
toa_s = T("get_mjds x68", lambda: [p.toas.get_mjds().value.astype(float)*86400 for p in psrs])
counts=[len(t) for t in toa_s]
up = T("upload_packed", lambda: dv.upload_packed([C**0.5, w, np.concatenate(toa_s), grid["ut"], np.repeat(np.arange(68,dtype=np.int32),counts)]))
sqrtC,w_d,toa_d,ut_d,psr_of = up
npts=grid["npts"]; ldt=rn.pad16(npts); dt=grid["dt"]; s=dv.stream_ptr()
Tm = dv.empty((2*(Nf-2), ldt))
T("alloc T", lambda: dv.empty((2*(Nf-2), ldt)))
T("twiddle", lambda: _lib.call("pta_gwb_twiddle", dv.ptr(sqrtC), Nf, npts, 10, ctypes.c_double(1.0/dt), dv.ptr(Tm), ldt, s))
G0=dv.empty((68,npts)); G=dv.empty((68,npts))
T("idft", lambda: _lib.call("pta_gwb_idft", dv.ptr(w_d), 2*Nf, 68, Nf, dv.ptr(Tm), ldt, npts, dv.ptr(G0), npts, 1, s))
T("mix", lambda: _lib.call("pta_gwb_mix", dv.ptr(M), 68, dv.ptr(G0), 1, npts, npts, dv.ptr(G), 0, s))
ntot=sum(counts); jlo=dv.empty((ntot,),dtype=torch.int32); out=dv.empty((1,ntot))
T("bracket", lambda: _lib.call("pta_gwb_bracket", dv.ptr(ut_d), npts, dv.ptr(toa_d), ntot, dv.ptr(jlo), s))
T("interp", lambda: _lib.call("pta_gwb_interp", dv.ptr(G), npts, 68, npts, dv.ptr(ut_d), dv.ptr(toa_d), dv.ptr(psr_of), dv.ptr(jlo), ntot, 1, ctypes.c_double(1.0), dv.ptr(out), ntot, 0, s))
res = T("download", lambda: dv.download(out[0]))
from pta_replicator_amd._compat import TimeDelta, u
res_gw=np.split(res, np.cumsum(counts)[:-1])
def book():
    for p in psrs: make_ideal(p)
    t0=time.perf_counter()
    for ct,p in enumerate(psrs):
        dt_=res_gw[ct]/86400.0*u.day
        p.toas.adjust_TOAs(TimeDelta(dt_.to("day")))
        p.update_added_signals("{}_gwb".format(p.name), {"amplitude":1,"spectral_index":2}, dt_)
        p.update_residuals()
    return time.perf_counter()-t0
book(); print("bookkeeping x68            ", round(min(book() for _ in range(5))*1e3,3), "ms")
