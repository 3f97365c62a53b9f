import sys
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/gaussian-pcloud-render_amd']
import numpy as np, util
import test_gpu_fuzz as F
from oracle.oracle import Oracle
f=np.float32; d=np.float64
def fma(a,b,c): return (a.astype(d)*b.astype(d)+c.astype(d)).astype(f)
case=int(sys.argv[1]); gid=int(sys.argv[2])
s,mode=F._case(case)
dL=util.seeded_dL(s,seed=77+case)
o,g=Oracle().forward_backward(s,dL,exact=True)
ex=g["exact"]
W,H=s.W,s.H
m2=o["means2D"].astype(f); co=o["conic_opacity"].astype(f)
rgb=(o["rgb"] if s.colors_precomp is None else s.colors_precomp).astype(f)
vals=o["vals"]; ranges=o["ranges"]; nc=o["n_contrib"]; fT=o["final_T"].astype(f)
gx=(W+15)//16; gy=(H+15)//16
bg=s.bg.astype(f)
# per-pixel q for gaussian gid, both formulations
qlib=np.zeros((H,W),f); hit=np.zeros((H,W),bool)
refterm=np.zeros((H,W,3),f)  # conic x,y,w terms of the reference (float)
refmean=np.zeros((H,W,2),f)
for ty in range(gy):
  for tx in range(gx):
    tile=ty*gx+tx; r0,r1=ranges[tile]
    if r1<=r0: continue
    ys=np.arange(ty*16,min(ty*16+16,H)); xs=np.arange(tx*16,min(tx*16+16,W))
    PX,PY=np.meshgrid(xs,ys); PX=PX.ravel(); PY=PY.ravel()
    pxf=PX.astype(f); pyf=PY.astype(f)
    last=nc[PY,PX].astype(np.int64)
    Tfin=fT[PY,PX]; T=Tfin.copy(); Tl=Tfin.copy()
    dpx=[dL[c,PY,PX].astype(f) for c in range(3)]
    bgdot=np.zeros_like(T)
    for c in range(3): bgdot=(bgdot+(bg[c]*dpx[c]).astype(f)).astype(f)
    acc=[np.zeros_like(T) for _ in range(3)]; lastc=[np.zeros_like(T) for _ in range(3)]; la=np.zeros_like(T)
    s_rec=np.zeros_like(T); last_d=np.zeros_like(T); la_l=np.zeros_like(T)
    for k in range(r1-r0-1,-1,-1):
        i=vals[r0+k]
        act=(k<last)
        if not act.any(): continue
        dx=(m2[i,0]-pxf).astype(f); dy=(m2[i,1]-pyf).astype(f)
        A,B,C,op=co[i]
        power=((f(-0.5)*(((A*dx).astype(f)*dx).astype(f)+((C*dy).astype(f)*dy).astype(f)).astype(f)).astype(f)-((B*dx).astype(f)*dy).astype(f)).astype(f)
        G=np.exp(power.astype(d)).astype(f)
        alpha=np.minimum(f(0.99),(op*G).astype(f))
        h=act&~(power>0)&~(alpha<f(1/255))
        if not h.any(): continue
        a=np.where(h,alpha,f(0))
        # reference
        Tn_ref=(T/(f(1)-a)).astype(f)
        dLa=np.zeros_like(T)
        nacc=[None]*3
        for c in range(3):
            nacc[c]=((la*lastc[c]).astype(f)+((f(1)-la).astype(f)*acc[c]).astype(f)).astype(f)
            dLa=(dLa+((rgb[i,c]-nacc[c]).astype(f)*dpx[c]).astype(f)).astype(f)
        dLa=(dLa*Tn_ref).astype(f)
        dLa=(dLa+(((-Tfin/(f(1)-a)).astype(f))*bgdot).astype(f)).astype(f)
        for c in range(3):
            acc[c]=np.where(h,nacc[c],acc[c]); lastc[c]=np.where(h,rgb[i,c],lastc[c])
        la=np.where(h,a,la); T=np.where(h,Tn_ref,T)
        # lib
        om=(f(1)-a).astype(f); rcp=(f(1)/om).astype(f); q0=(Tl*rcp).astype(f); Tn=fma(fma(-om,q0,Tl),rcp,q0)
        dd=fma(np.full_like(T,rgb[i,2]),dpx[2],fma(np.full_like(T,rgb[i,1]),dpx[1],(rgb[i,0]*dpx[0]).astype(f)))
        sn=fma(la_l,(last_d-s_rec).astype(f),s_rec)
        dl=((dd-sn).astype(f)*Tn).astype(f)
        dl=fma((-Tfin*rcp).astype(f),bgdot,dl)
        Tl=Tn; s_rec=sn; last_d=dd; la_l=a   # (non-hit lanes: a=0 -> identity, as in the kernel)
        if i==gid:
            qv=np.where(h,(G*dl).astype(f),f(0))
            qlib[PY,PX]=qv; hit[PY,PX]=h
            dL_dG=(op*dLa).astype(f); gdx=(G*dx).astype(f); gdy=(G*dy).astype(f)
            t0=((((f(-0.5)*gdx).astype(f))*dx).astype(f)*dL_dG).astype(f)
            t1=((((f(-0.5)*gdx).astype(f))*dy).astype(f)*dL_dG).astype(f)
            t3=((((f(-0.5)*gdy).astype(f))*dy).astype(f)*dL_dG).astype(f)
            for j,t in enumerate((t0,t1,t3)): refterm[PY,PX,j]=np.where(h,t,f(0))
            dGx=((-gdx*A).astype(f)-(gdy*B).astype(f)).astype(f); dGy=((-gdy*C).astype(f)-(gdx*B).astype(f)).astype(f)
            refmean[PY,PX,0]=np.where(h,((dL_dG*dGx).astype(f)*f(0.5*W)).astype(f),f(0))
            refmean[PY,PX,1]=np.where(h,((dL_dG*dGy).astype(f)*f(0.5*H)).astype(f),f(0))
A,B,C,op=[d(x) for x in co[gid]]
mx,my=d(m2[gid,0]),d(m2[gid,1])
exc=ex["dL_dconic"][gid][[0,1,3]]; exm=ex["dL_dmean2D"][gid][:2]
print("hit pixels",hit.sum(),"exact conic",exc,"mean",exm)
# (a) exact double from lib float q
YY,XX=np.mgrid[0:H,0:W]
ddx=mx-XX; ddy=my-YY
qa=qlib.astype(d)
ca=-0.5*op*np.array([(qa*ddx*ddx).sum(),(qa*ddx*ddy).sum(),(qa*ddy*ddy).sum()])
ma=op*np.array([0.5*W*-(qa*(A*ddx+B*ddy)).sum(), 0.5*H*-(qa*(C*ddy+B*ddx)).sum()])
print("(a) lib q, exact sums: conic rel err",(ca-exc)/np.abs(exc),"mean rel err",(ma-exm)/np.abs(exm))
# (c) reference float terms, exact sums
cc=refterm.astype(d).sum((0,1)); mc=refmean.astype(d).sum((0,1))
print("(c) ref terms, exact sums: conic rel err",(cc-exc)/np.abs(exc),"mean rel err",(mc-exm)/np.abs(exm))
# (c2) ref terms, float sequential sum (one order)
sc=np.zeros(3,f)
for t in refterm.reshape(-1,3):
    if t.any(): sc=(sc+t).astype(f)
print("(c2) ref terms, float sequential: conic rel err",(sc.astype(d)-exc)/np.abs(exc))
# (b) lib moments per quadrant, float sequential fma in 2x2-block order, float shift, double sum over quadrants
cb=np.zeros(3); mb=np.zeros(2); cb_f=np.zeros(3,f)
for y0 in range(0,H,8):
  for x0 in range(0,W,8):
    blk=qlib[y0:y0+8,x0:x0+8]
    if not blk.any(): continue
    q64=np.zeros((8,8),f); q64[:blk.shape[0],:blk.shape[1]]=blk
    S=np.zeros(6,f)
    order=[(2*m+(k>>1),2*r+(k&1)) for m in range(4) for r in range(4) for k in range(4)]
    for (yy,xx) in order:
        cx=f(xx-3.5); cy=f(yy-3.5); q=q64[yy,xx]
        for j,b in enumerate((f(1),cx,cy,f(cx*cx),f(cx*cy),f(cy*cy))):
            S[j]=f(d(b)*d(q)+d(S[j]))
    S1,Sx,Sy,Sxx,Sxy,Syy=S
    bx=f(f(m2[gid,0])-f(x0+3.5)); by=f(f(m2[gid,1])-f(y0+3.5))
    Dx=f(d(bx)*d(S1)-d(Sx)); Dy=f(d(by)*d(S1)-d(Sy))
    t2x=f(d(bx)*d(Dx)+d(f(d(-bx)*d(Sx)+d(Sxx))))
    t2y=f(d(by)*d(Dx)+d(f(d(-bx)*d(Sy)+d(Sxy))))
    t2w=f(d(by)*d(Dy)+d(f(d(-by)*d(Sy)+d(Syy))))
    cO=f(co[gid,3])
    o=np.array([f(f(f(-0.5)*cO)*t) for t in (t2x,t2y,t2w)],f)
    cb+=o.astype(d); cb_f=(cb_f+o).astype(f)
    m_x=f(f(cO*f(-0.5*W))*f(d(co[gid,0])*d(Dx)+d(f(co[gid,1]*Dy))))
    m_y=f(f(cO*f(-0.5*H))*f(d(co[gid,1])*d(Dx)+d(f(co[gid,2]*Dy))))
    mb+=np.array([m_x,m_y],d)
print("(b) lib moments float, double sum of quadrants: conic rel err",(cb-exc)/np.abs(exc),"mean rel err",(mb-exm)/np.abs(exm))
print("(b2) ... float sequential sum of quadrants: conic rel err",(cb_f.astype(d)-exc)/np.abs(exc))
