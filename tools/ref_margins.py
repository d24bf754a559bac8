import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import bioen_amd
from oracle import ref_binding as R
from oracle import cpus
from test_hip_fullsize import _targets, LBFGS_DEFAULTS
R.set_fast_openmp_flag(0); R.omp_set_num_threads(cpus.usable_cpus())
for prior, M, N in [("uniform",256,100000),("random",256,100000),("uniform",1024,20000),("uniform",205,50000)]:
    conv = dict(LBFGS_DEFAULTS, epsilon=1e-9, delta=0.0, past=0, max_iterations=200000)
    thetas = [316.0,100.0,31.6] if (M,N)==(256,100000) else [316.0,100.0]
    YTrue, sig_sim, sig_exp, YTilde = _targets(M)
    if prior=="uniform": G=np.zeros(N); g0=G
    else:
        thetas=thetas[:2]; conv=dict(conv, epsilon=1e-10)
        rng=np.random.default_rng(99); G=np.log(rng.gamma(2.0,1.0,N)); G-=G.max(); g0=G+0.3*rng.standard_normal(N)
    with bioen_amd.Context.synthetic(M,N,YTrue,sig_sim,sig_exp,YTilde,seed=12345) as ctx:
        res,w,infos=ctx.opt_lbfgs_logw_batch(thetas,g0,G,conv); yT=np.ascontiguousarray(ctx.read_ytilde())
    for k,th in enumerate(thetas):
        g_ref,fmin_ref,code=R.opt_lbfgs_logw(g0,G,yT,YTilde,th,conv)
        w_ref=np.asarray(R.get_weights(g_ref)[0]).ravel()
        print("logw",prior,M,N,th,code,infos[k].lbfgs_code,"fmin rel %.2e"%(abs(infos[k].fmin-fmin_ref)/abs(fmin_ref)),"w %.2e"%(np.abs(w[k]-w_ref).max()/w_ref.max()), infos[k].iterations)
for M,N in [(256,100000),(512,50000),(96,30000)]:
    conv = dict(LBFGS_DEFAULTS, epsilon=1e-9, delta=0.0, past=0, max_iterations=200000)
    thetas=[316.0,100.0,31.6]
    YTrue, sig_sim, sig_exp, YTilde = _targets(M); w0=np.full(N,1.0/N)
    with bioen_amd.Context.synthetic(M,N,YTrue,sig_sim,sig_exp,YTilde,seed=12345) as ctx:
        res,w,infos=ctx.opt_lbfgs_forces_batch(thetas,np.zeros(M),w0,conv); yT=np.ascontiguousarray(ctx.read_ytilde())
    for k,th in enumerate(thetas):
        f_ref,fmin_ref,code=R.opt_lbfgs_forces(np.zeros(M),w0,yT,YTilde,th,conv)
        w_ref=np.asarray(R.forces_weights(f_ref,w0,yT)).ravel()
        print("forces",M,N,th,code,infos[k].lbfgs_code,"fmin rel %.2e"%(abs(infos[k].fmin-fmin_ref)/abs(fmin_ref)),"w %.2e"%(np.abs(w[k]-w_ref).max()/w_ref.max()))
