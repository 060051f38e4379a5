import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from ml_amd import _lib, synth
from oracle import oracle_ctypes as orc
d=K=16; n=100000
mix=synth.Mixture(d,K,diagonal=True)
X,_=mix.sample(n)
var0=np.tile(np.var(X,axis=0),(K,1)); pi0=np.full(K,1/K); mu0=mix.initial_means()
ctx=_lib.Context()
dt=_lib.Data(ctx,X)
em=orc.EM(K); em.set_covariance_type("diag")
pi,mu,var=pi0,mu0,var0
for it in range(3):
    em.set_parameters(mu, np.stack([np.diag(v) for v in var]), pi)
    em.expectation_step(X)
    R0=em.responsibilities.copy(); ll0=em.log_likelihood
    out={}
    for name,env in (("exact",{"MLHIP_DIAG_GEMM":"0"}),("gemm",{"MLHIP_DIAG_GEMM":"1","MLHIP_DIAG_EXPAND_LIMIT":"1e9"}),("mixed",{"MLHIP_DIAG_MIXED":"1"})):
        for k in ("MLHIP_DIAG_GEMM","MLHIP_DIAG_EXPAND_LIMIT","MLHIP_DIAG_MIXED"): os.environ.pop(k,None)
        os.environ.update(env)
        ll,pi1,mu1,var1=dt.em_step_diag(pi,mu,var)
        R=dt.em_responsibilities(K)
        out[name]=(abs(ll-ll0)/abs(ll0), np.max(np.abs(R-R0)))
    s=(X.mean(0)); b2=(((mu-s)**2)/var).sum(1)
    print(it, "maxB2=%.0f"%b2.max(), {k:("%.2e"%v[0],"%.2e"%v[1]) for k,v in out.items()})
    em.maximisation_step(X)
    pi,mu=em.mixing_probabilities.copy(),em.means.copy(); var=np.array([np.diag(c) for c in em.covariances])
