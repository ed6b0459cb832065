#!/usr/bin/env python3
"""NumPy prototype of the parallel-in-time propagation of a long gap (vinsat_amd/csrc/vba_long.hip): convergence table of the
parareal iteration (fine = the reference's chain of 1 s RK4 steps, coarse = one RK4 step per chunk) against the serial walk."""
import numpy as np, math
MU=398600.4418; J2C=1.75553e10
def accel(p):
    px2,py2,pz2=p[0]*p[0],p[1]*p[1],p[2]*p[2]
    r2=px2+py2+pz2; r=math.sqrt(r2); r3=r*r*r; r7=r3*r3*r
    k3=MU/r3; k7=J2C/r7
    u0=6*px2-1.5*py2-1.5*pz2; u2=3*px2-4.5*py2-4.5*pz2
    return np.array([-k3*p[0]+k7*u0*p[0], -k3*p[1]+k7*u0*p[1], -k3*p[2]+k7*u2*p[2]])
def f(x): return np.concatenate([x[3:],accel(x[:3])])
def rk4(x,h):
    k1=f(x);k2=f(x+0.5*h*k1);k3=f(x+0.5*h*k2);k4=f(x+h*k3)
    return x+(h/6)*(k1+2*k2+2*k3+k4)
def fine(x,n):
    for _ in range(n): x=rk4(x,1.0)
    return x
def parareal(x0,s,tol=2.0**-40,verbose=False):
    L=math.ceil(math.sqrt(s)) if s<=1024 else math.ceil(s/32); P=math.ceil(s/L)
    lens=[L]*(P-1)+[s-(P-1)*L]
    U=[x0]; Gold=[]
    for j in range(P):
        g=rk4(U[j],float(lens[j])); Gold.append(g); U.append(g)
    for k in range(1,P+1):
        F=[fine(U[j],lens[j]) for j in range(P)]
        Un=[x0]; d=0.0
        for j in range(P):
            gn=rk4(Un[j],float(lens[j]))
            un=F[j]+(gn-Gold[j]); Gold[j]=gn
            sc=np.array([np.abs(un[:3]).max()]*3+[np.abs(un[3:]).max()]*3)
            d=max(d,(np.abs(un-U[j+1])/sc).max())
            Un.append(un)
        U=Un
        if verbose: print(' iter',k,'delta',d)
        if d<=tol: break
    return U[-1],k,P,L
a=6978.0
x0=np.array([-a*0.6, 100.0, a*0.8, 0.5, -7.5, 0.3]); 
v=math.sqrt(MU/a); 
# polar-ish orbit
p=np.array([-a*0.6,0.0,a*0.8]); vd=np.array([0.8,0.0,0.6])*v*1.001
x0=np.concatenate([p,vd])
for s in (65,100,300,945,1000,3000,6000):
    ref=fine(x0,s)
    xp,k,P,L=parareal(x0,s,verbose=(s in(945,6000)))
    print(s,'P',P,'L',L,'iters',k,'rel err pos',np.abs(xp[:3]-ref[:3]).max()/np.abs(ref[:3]).max(),'vel',np.abs(xp[3:]-ref[3:]).max()/np.abs(ref[3:]).max())
# perturbed (bad initial guess, 100 km off, 10% v)
x1=x0.copy(); x1[:3]+=np.array([60,-50,40.0]); x1[3:]*=1.1
for s in (945,):
    ref=fine(x1,s); xp,k,P,L=parareal(x1,s,verbose=True)
    print(s,'iters',k,'rel err',np.abs(xp[:3]-ref[:3]).max()/np.abs(ref[:3]).max(),np.abs(xp[3:]-ref[3:]).max()/np.abs(ref[3:]).max())
