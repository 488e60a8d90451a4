"""Dev tool (CPU, 8 processes): the headline setting in the oracle on 40 families of 64 paths — KP (N 120 / 200 / 233), KPC, K; corridors scaled by 1 / 0.7 / 0.5 / 0.35; start offsets x 1 / 3 —
solved, certified and iteration counts per family (round 4: every solved path certified, longest 143 iterations; unsolvable paths are infeasible corridors)."""
import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, multiprocessing as mp
def work(a):
    form, cfg, first, scale, x0s, N = a
    from oracle import oracle_py as O
    from path_optimizer_amd import synth
    kw={}
    if form==2: kw["formulation"]=2
    if N: kw.update(N=N)
    b = synth.make_batch(cfg, B=64, first_path=first, **kw)
    b.bounds = b.bounds*scale
    b.x0 = b.x0.copy(); b.x0[:,0]*=x0s
    p=O.device_equivalent_params()
    for k,v in dict(refine=2, refine_rounds=5, refine_extra_rounds=2, refine_eps=1e-8).items(): setattr(p,k,v)
    _,info,_=O.solve_batch(b,p)
    s=info["status"]==1
    return (form,cfg,first,scale,x0s,N,int(s.sum()),int(((info["status_refine"]==1)&s).sum()),float(info["iters"][s].mean()) if s.any() else 0.0,int(info["iters"][s].max()) if s.any() else 0, int((info["status"]==-3).sum()))
if __name__=="__main__":
    jobs=[]
    rng=np.random.default_rng(3)
    for form,cfg in ((0,3),(1,5),(2,3)):
        for scale in (1.0,0.7,0.5,0.35):
            for x0s in (1.0,3.0):
                for N in (None,) if form!=0 else (None,120,233):
                    jobs.append((form,cfg,int(rng.integers(0,4000)),scale,x0s,N))
    with mp.get_context("spawn").Pool(8) as pool:
        for r in pool.imap_unordered(work,jobs):
            form,cfg,first,scale,x0s,N,ns,nc,mean,mx,ninf=r
            flag = "" if nc==ns else "   <-- uncertified %d"%(ns-nc)
            print(f"form {form} N {N} scale {scale} x0x{x0s}: solved {ns}/64 (infeasible {ninf}) certified {nc} iters mean {mean:.1f} max {mx}{flag}", flush=True)
