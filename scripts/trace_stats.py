#!/usr/bin/env python
"""Algorithmic statistics of a scheduling cycle from the ORACLE's decision trace (no GPU): shapes per window, same-shape run lengths,
how often a row is won by a node its window already changed.  Usage: python scripts/trace_stats.py [config index, default 3]  """
import sys, time; import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import importlib, numpy as np
kbm=importlib.import_module("kube-batch_amd")
import oracle as om
cfgi=int(sys.argv[1]) if len(sys.argv)>1 else 3
conf=kbm.conf.load_scheduler_conf()
if cfgi==4:
    import bench; conf=kbm.conf.load_scheduler_conf(bench.BINPACK_CONF)   # configs[3]: binpack weights
snap=kbm.snapshot.synth(kbm.snapshot.synth_config(cfgi,1.0))
t0=time.time(); o=om.Oracle(conf,snap,threads=16); o.allocate(); print('oracle allocate s',round(time.time()-t0,1))
dec=o.decisions(); print('decisions',len(dec),'popped',o.popped)
T=snap.n_tasks
# row shape id = (init vector, nz, class)
key=np.concatenate([snap.task_init_resreq.T, snap.task_nz_cpu[:,None].astype(float), snap.task_nz_mem[:,None].astype(float), snap.task_class[:,None].astype(float)],axis=1)
_,shape=np.unique(key,axis=0,return_inverse=True)
print('distinct shapes overall',shape.max()+1)
for W in (128,256,512):
    nwin=0; dirty_win=0; tot=0; shapes_per=[]; runs=[]; dirty_rows_after_first=0; chains=[]
    for a in range(0,len(dec),W):
        w=dec[a:a+W]; nwin+=1
        seen=set(); sh=shape[w[:,0]]
        shapes_per.append(len(set(sh.tolist())))
        # same-shape runs
        r=1
        for i in range(1,len(sh)):
            if sh[i]==sh[i-1]: r+=1
            else: runs.append(r); r=1
        runs.append(r)
        c=0
        for (t,n,k) in w:
            if n in seen: dirty_win+=1; c+=1
            else:
                if c: chains.append(c)
                c=0
            seen.add(n); tot+=1
        if c: chains.append(c)
    runs=np.array(runs); chains=np.array(chains) if chains else np.array([0])
    print(f'W={W}: windows {nwin}, dirty-winner rows {dirty_win}/{tot} = {dirty_win/tot:.3f}, shapes/window mean {np.mean(shapes_per):.1f} max {max(shapes_per)}, same-shape run mean {runs.mean():.2f} median {np.median(runs)} p90 {np.percentile(runs,90)}, dirty chains: n {len(chains)} mean {chains.mean():.2f} max {chains.max()}')
# how often is the winner the SAME node as the previous row's
same_prev=(dec[1:,1]==dec[:-1,1]).mean(); print('winner == previous row winner:',round(float(same_prev),3))
