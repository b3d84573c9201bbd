"""TEST INFRASTRUCTURE ONLY -- round-5 experiment #6 (profiles/r05/experiments/README.md): Gauss-Newton multiple shooting with a LINEAR
forward sweep instead of the nonlinear line-search rollout, on the numpy oracle's model code.  Not shipped, not imported by the product."""
import numpy as np, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cport, ddp as oddp, models as omodels
from srbd_horizon_amd import workload

def solve_gnms(model, x0, P, xs_ws, us_ws, opt, max_ls=40):
    us=np.array(us_ws,float); N=us.shape[0]
    xs=np.array(xs_ws,float); xs[0]=x0
    d=oddp.defects(model,xs,us,P); J=oddp.total_cost(model,xs,us,P); gap=float(np.sum(np.abs(d)))
    mu=opt.mu0; rho=0.0; theta=0.0; iters=0; status=1; conv=False; evals=0
    while iters<opt.max_iters:
        while True:
            ok,K,kff,dV1,dV2,G1,G2,_,_,_=oddp.backward_pass(model,xs,us,P,d,mu,theta,int(opt.second_order))
            if ok: break
            if theta: theta=0.0; continue
            mu=max(mu,0.0)*10+opt.mu_min
            if mu>opt.mu_max: return iters,False,2,J,xs,us,evals
        expected=-(dV1+dV2)
        if expected<opt.cost_reduction_ths and gap<=opt.gap_tol: conv,status=True,0; break
        A1=dV1+G1; B2=dV2+G2
        if gap>0: rho=max(rho,2*max(A1,A1+B2,0.0)/gap)
        # linear forward sweep: full Newton step of the linearised problem
        dx=np.zeros_like(xs); du=np.zeros_like(us)
        for k in range(N):
            du[k]=kff[k]+K[k]@dx[k]
            fx,fu=model.f_jac(xs[k],us[k],P[k])
            dx[k+1]=fx@dx[k]+fu@du[k]+d[k]
        a=opt.alpha_0; acc=False; slack=1e-13*(abs(J)+rho*gap)
        while a>=opt.alpha_converge_threshold:
            xn=xs+a*dx; un=us+a*du
            dn=oddp.defects(model,xn,un,P); Jn=oddp.total_cost(model,xn,un,P); gn=float(np.sum(np.abs(dn))); evals+=1
            pred=a*A1+a*a*B2-a*rho*gap
            dphi=(Jn+rho*gn)-(J+rho*gap)
            if np.isfinite(Jn) and dphi<=opt.beta*pred+slack: acc=True; break
            a*=opt.line_search_decrease_factor
        if not acc:
            if theta: theta=0.0; continue
            conv=bool(gap<=opt.gap_tol and expected<=opt.cost_reduction_ths*max(1,abs(J))); status=0 if conv else 4; break
        theta=1.0 if (opt.second_order and a==opt.alpha_0) else 0.0
        dJ=J-Jn; xs,us,J,d,gap=xn,un,Jn,dn,gn; iters+=1
        if mu>opt.mu0: mu=max(opt.mu0,mu*0.1)
        if abs(dJ)<opt.cost_reduction_ths and gap<=opt.gap_tol: conv,status=True,0; break
    return iters,conv,status,J,xs,us,evals

if __name__=="__main__":
    N=30; seeds=np.arange(int(sys.argv[1]) if len(sys.argv)>1 else 120)
    batch=workload.make_batch("srbd13",N,seeds)
    cst=omodels.RobotConsts(**batch["consts"]); m=omodels.make_model("srbd13",cst)
    OPTS=dict(max_iters=100,alpha_converge_threshold=1e-12,beta=1e-3)
    gap_tol=float(sys.argv[2]) if len(sys.argv)>2 else 1e-9
    o=oddp.DdpOptions(**dict(OPTS,gap_tol=gap_tol))
    xo,uo,so=cport.solve_batch(cst,oddp.DdpOptions(**OPTS),batch["x0"],batch["params"],batch["xs"],batch["us"],threads=8)
    t=time.time(); res=[]
    for b in range(len(seeds)):
        it,cv,st,J,xs,us,ev=solve_gnms(m,batch["x0"][b],batch["params"][b],batch["xs"][b],batch["us"][b],o)
        res.append((it,cv,st,J,ev,np.max(np.abs(xs-xo[b])),abs(J-so[b,0])/so[b,0]))
    r=np.array([(a,b,c,e) for a,b,c,_,e,_,_ in res],float)
    print("time",time.time()-t)
    print("DDP  : mean iters",so[:,1].mean(),"max",so[:,1].max(),"conv",so[:,2].mean())
    print("GNMS : mean iters",r[:,0].mean(),"max",r[:,0].max(),"conv",r[:,1].mean(),"status",np.unique(r[:,2],return_counts=True),"merit evals per iter",r[:,3].sum()/max(r[:,0].sum(),1))
    both=(so[:,2]==1)&(r[:,1]==1)
    print("same optimum: max linf x",max(x[5] for x,bb in zip(res,both) if bb),"max rel cost",max(x[6] for x,bb in zip(res,both) if bb))
    print("iters DDP vs GNMS (first 30):",list(zip(so[:30,1].astype(int),r[:30,0].astype(int))))
