import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nonstationary_multivariate_gaussian_process_amd import _lib, sim
ctx=_lib.Context(0)
for (N,M) in [(1024,5),(2048,5),(4096,5)]:
    d=sim.simulate_separable(N,M,5)
    pars=sim.perturb(d["pars_true"],0.05,0.4)
    hv=[sim.HYPER_SEP[k] for k in ["mu_tilde_l","alpha_tilde_l","beta_tilde_l","mu_tilde_sigma","alpha_tilde_sigma","beta_tilde_sigma","a","b","c"]]
    ctx.set_data(d["x"],d["Y"])
    for _ in range(2):   # warm both modes twice (scratch sizes, cached prior factors, rocBLAS kernel loading)
        ctx.logpos_sep(pars,hv,True,True); ctx.logpos_sep(pars,hv,True,False)
    ctx.profile_enable(True); ctx.profile_reset()
    t0=time.perf_counter(); 
    for _ in range(3): out,_=ctx.logpos_sep(pars,hv,True,False)
    t1=time.perf_counter()
    for _ in range(3): out,g=ctx.logpos_sep(pars,hv,True,True)
    t2=time.perf_counter()
    pr=ctx.profile_read(); ctx.profile_enable(False)
    print(N,M,"value %.1f ms  value+grad %.1f ms"%((t1-t0)/3*1e3,(t2-t1)/3*1e3), {k:round(v[0]/max(v[1],1),3) for k,v in pr.items() if v[1]}, out[:2])
