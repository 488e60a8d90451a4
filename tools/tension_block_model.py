"""Dev tool (CPU): residual / forward error of the blocked band substitution of the TENSION smoothing QP in its variants — plain substitution, explicit S_k = T^-T D^-1 T^-1
(round 3), the same with exactly rounded entries (SX), and the factored forms Wm' (Wm r) with Wm = D^-1/2 T^-1 (W: what po_smooth.hip band_solve_blocks applies since round 4) and
T^-T (D^-1 (T^-1 r)) (TDT) — against a long-double solve, on the reduced KKT matrices of oracle-assembled TENSION QPs (RCM-ordered to half-bandwidth 9)."""
import sys
sys.path.insert(0,'/root/repo')
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
from oracle import oracle_py as O
from path_optimizer_amd import synth
kind=1
d = synth.make_distance_map(3); om = O.make_map(*d[:4])
inp = synth.make_smooth_inputs(22, 8, P=100, kind=kind, ragged=False, jitter_ds=True)
LD = np.longdouble
def ldl_band(M):
    n=M.shape[0]; L=np.eye(n); D=np.zeros(n); A=M.copy()
    for j in range(n):
        D[j]=A[j,j]
        L[j+1:,j]=A[j+1:,j]/D[j]
        A[j+1:,j+1:]-=np.outer(L[j+1:,j],L[j+1:,j])*D[j]
    return L,D
def run(b, rho, sigma=1e-6):
    P,q,A,l,u = O.smooth_assemble(kind, O.default_params(), inp, b=b, m_map=om)
    P=np.asarray(P.todense() if sp.issparse(P) else P, dtype=float); A=np.asarray(A.todense() if sp.issparse(A) else A, dtype=float)
    P=np.triu(P)+np.triu(P,1).T
    eq = np.abs(u-l)<1e-4
    rv = np.where(eq, 1e3*rho, rho)
    # crude Ruiz scaling of the KKT-ish: scale variables by D = 1/sqrt(diag(M))
    M = P + sigma*np.eye(P.shape[0]) + A.T@(rv[:,None]*A)
    perm = reverse_cuthill_mckee(sp.csr_matrix(np.abs(M)>0), symmetric_mode=True)
    M = M[np.ix_(perm,perm)]
    n=M.shape[0]
    bw = max(abs(i-j) for i,j in zip(*np.nonzero(M)))
    L,D = ldl_band(M)
    W=9; nb=(n+W-1)//W; npad=nb*W
    Lp=np.eye(npad); Lp[:n,:n]=L; Dp=np.ones(npad); Dp[:n]=D
    rng=np.random.default_rng(b)
    xt = rng.standard_normal(n)
    bvec = (M.astype(LD)@xt.astype(LD)).astype(float)   # rhs with known (approx) solution
    # reference solve in long double
    Lq,Dq = Lp.astype(LD), Dp.astype(LD)
    bp=np.zeros(npad); bp[:n]=bvec
    def subst(Lm,Dm,rhs,dt):
        y=rhs.astype(dt).copy()
        for j in range(npad):
            y[j+1:min(npad,j+bw+1)] -= Lm[j+1:min(npad,j+bw+1),j]*y[j]
        y=y/Dm
        for j in range(npad-1,-1,-1):
            y[max(0,j-bw):j] -= Lm[j,max(0,j-bw):j]*y[j]
        return y
    xref = subst(Lq,Dq,bp,LD)
    xs = subst(Lp,Dp,bp,float)
    # blocked with explicit inverses (float64)
    T=[Lp[k*W:(k+1)*W,k*W:(k+1)*W] for k in range(nb)]
    C=[Lp[(k+1)*W:(k+2)*W,k*W:(k+1)*W] for k in range(nb-1)]
    Ti=[np.linalg.solve(t,np.eye(W)) for t in T]   # (device: column-wise forward substitution)
    def tinv(t):
        X=np.zeros((W,W))
        for c in range(W):
            col=np.zeros(W)
            for i in range(W):
                acc=0.0
                for j in range(i): acc-=t[i,j]*col[j]
                col[i]=1.0 if i==c else (0.0 if i<c else acc)
            X[:,c]=col
        return X
    Ti=[tinv(t) for t in T]
    Mk=[C[k]@Ti[k] for k in range(nb-1)]
    Sk=[Ti[k].T@np.diag(1.0/Dp[k*W:(k+1)*W])@Ti[k] for k in range(nb)]
    def tinv_ld(t):
        return np.linalg.inv(t.astype(np.float64)).astype(LD)  # placeholder, refined below
    TiL=[]
    for t in T:
        X=np.zeros((W,W),dtype=LD); tl=t.astype(LD)
        for c in range(W):
            col=np.zeros(W,dtype=LD)
            for i in range(W):
                acc=LD(0)
                for j in range(i): acc-=tl[i,j]*col[j]
                col[i]=LD(1) if i==c else (LD(0) if i<c else acc)
            X[:,c]=col
        TiL.append(X)
    SkX=[(TiL[k].T@np.diag(LD(1)/Dp[k*W:(k+1)*W].astype(LD))@TiL[k]).astype(float) for k in range(nb)]
    MkX=[(C[k].astype(LD)@TiL[k]).astype(float) for k in range(nb-1)]
    Wk=[np.diag(1.0/np.sqrt(Dp[k*W:(k+1)*W]))@Ti[k] for k in range(nb)]
    def blocked(form):
        r=[None]*nb; r[0]=bp[:W].copy()
        for k in range(nb-1): r[k+1]=bp[(k+1)*W:(k+2)*W]-Mk[k]@r[k]
        if form=="S": g=[Sk[k]@r[k] for k in range(nb)]
        elif form=="SX": g=[SkX[k]@r[k] for k in range(nb)]
        elif form=="W": g=[Wk[k].T@(Wk[k]@r[k]) for k in range(nb)]
        elif form=="TDT": g=[Ti[k].T@((Ti[k]@r[k])/Dp[k*W:(k+1)*W]) for k in range(nb)]
        x=[None]*nb; x[nb-1]=g[nb-1]
        for k in range(nb-2,-1,-1): x[k]=g[k]-Mk[k].T@x[k+1]
        return np.concatenate(x)
    out={}
    for nm,x in (("subst",xs),("S",blocked("S")),("SX",blocked("SX")),("W",blocked("W")),("TDT",blocked("TDT"))):
        res = (bvec.astype(LD) - M.astype(LD)@x[:n].astype(LD)).astype(float)
        out[nm]=(np.abs(x-xref.astype(float)).max()/np.abs(xref).max(), np.abs(res).max()/np.abs(bvec).max())
    return n,bw,np.linalg.cond(M),out,max(np.linalg.cond(t) for t in T), max(np.abs(s).max() for s in Sk)
for rho in (0.1, 10.0, 1e3):
    for b in range(3):
        n,bw,cond,out,ct,smax=run(b,rho)
        print(f"rho {rho} b {b} n {n} bw {bw} cond {cond:.2e} condT {ct:.1e} |S|max {smax:.1e} | "+"  ".join(f"{k}: fwd {v[0]:.1e} res {v[1]:.1e}" for k,v in out.items()))
