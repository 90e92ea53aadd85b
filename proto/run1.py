import numpy as np, time, sys
import synth, qp, lcqp
for inst in range(int(sys.argv[1]) if len(sys.argv)>1 else 3):
    d = synth.gen(inst)
    t=time.time()
    log=[]
    r = lcqp.run_lcqp(d, lambda Q,A: qp.QPADMM(Q,A), log=log)
    print(inst, r['ret'], {k:v for k,v in r.items() if k not in('x','y','ret')}, 'time %.1f'%(time.time()-t))
    if r['ret']=='SUCCESS':
        x=r['x']; print('  phi', (d['L']@x)@(d['R']@x), 'obj', 0.5*x@d['Q']@x+d['g']@x)
