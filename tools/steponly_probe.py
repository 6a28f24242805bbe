"""Development tool: the step-only forms on long vectors (fa_reduce_step on 162.5 M elements x 32
clients; a 160-client FedAdam round on 86.5 M elements, whose first client chunk leaves a partial
step) under the default routing and with one kernel forced."""
import sys, torch
sys.path.insert(0, ".")
from appfl_amd import ops
def timed(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
n,K=162_500_000,32
rows=[torch.empty(n,device="cuda") for _ in range(K)]
for k,r in enumerate(rows): ops.synth_fill(r,100+k,scale=0.01)
g=torch.zeros(n,device="cuda"); step=torch.empty(n,device="cuda"); w=[1.0/K]*K
for name,st in (("default",2),("tile",0),("stash",-1),("default",2)):
    ops.tune(stash=st)
    t=timed(lambda: ops.reduce_step(rows,w,g,step))
    print(f"reduce_step n={n} K={K} {name}: {t:.3f} ms frac {4*(K+2)*n/t/8e9:.3f}",flush=True)
del rows
n,K=86_567_656,160
rows=[torch.empty(n,device="cuda") for _ in range(K)]
for k,r in enumerate(rows): ops.synth_fill(r,100+k,scale=0.01)
g=torch.zeros(n,device="cuda"); m=torch.zeros(n,device="cuda"); v=torch.full((n,),1e-6,device="cuda"); ws=torch.empty(n,device="cuda"); w=[1.0/K]*K
for name,st in (("default",2),("stash",-1),("default",2)):
    ops.tune(stash=st)
    t=timed(lambda: ops.reduce_opt(rows,w,g,m,v,"adam",lr=0.1,beta1=0.9,beta2=0.99,tau=1e-3,step_ws=ws))
    print(f"fedadam n={n} K={K} {name}: {t:.3f} ms frac {4*(K+8)*n/t/8e9:.3f}",flush=True)
