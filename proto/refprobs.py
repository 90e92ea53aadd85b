import numpy as np, os
def warm_up(x0=None, variant=None):
    d = dict(Q=2*np.eye(2), g=np.array([-2.,-2.]), L=np.array([[1.,0.]]), R=np.array([[0.,1.]]), A=np.zeros((0,2)), lbA=None, ubA=None, n=2, nC=0, nComp=1)
    if variant=='w_A':
        d.update(A=np.array([[1.,-1.]]), lbA=np.array([-0.5]), ubA=np.array([np.inf]), nC=1)
    if variant=='binary':
        d.update(L=np.array([[1.,0.],[1.,0.]]), R=np.array([[0.,1.],[-1.,0.]]), lbL=np.zeros(2), lbR=np.array([0.,-0.5]), nComp=2)
    return d
def circle(N=100):
    nV=2+2*N; nC=N+1; nComp=N
    Q=np.zeros((nV,nV)); Q[0,0]=Q[1,1]=17; Q[0,1]=Q[1,0]=-15
    for i in range(2,nV): Q[i,i]=5e-12
    xr=np.array([0.5,-0.6]); g=np.zeros(nV); g[:2]=-(np.array([[17,-15],[-15,17.]])@xr)
    A=np.zeros((nC,nV)); L=np.zeros((nComp,nV)); R=np.zeros((nComp,nV)); x0=np.zeros(nV); x0[:2]=xr
    for i in range(N):
        A[i,0]=np.cos(2*np.pi*i/N); A[i,1]=np.sin(2*np.pi*i/N); A[i,2+2*i]=1
        A[N,3+2*i]=1; L[i,2+2*i]=1; R[i,3+2*i]=1; x0[2*i+2]=1; x0[2*i+3]=1
    return dict(Q=Q,g=g,L=L,R=R,A=A,lbA=np.ones(nC),ubA=np.ones(nC),n=nV,nC=nC,nComp=nComp), x0
def example_data():
    p='/root/reference/examples/example_data/'
    ld=lambda f: np.loadtxt(p+f+'.txt')
    g=ld('g'); n=len(g); lbL=ld('lbL'); nComp=len(lbL); lbA=ld('lbA'); nC=len(lbA)
    d=dict(Q=ld('Q').reshape(n,n), g=g, L=ld('L').reshape(nComp,n), R=ld('R').reshape(nComp,n), A=ld('A').reshape(nC,n),
           lbA=lbA, ubA=ld('ubA'), lbL=lbL, ubL=ld('ubL'), lbR=ld('lbR'), ubR=ld('ubR'), n=n, nC=nC, nComp=nComp)
    return d, ld('x0'), ld('lb'), ld('ub')
