#!/usr/bin/env python3
"""Bank-conflict model of the exact-score stage of k_fast_cells (no GPU): for the survivors of the six-bit rejection test
on a synthetic KITTI-shaped frame, the LDS cycles of one ring-byte read of 64 consecutive work-list items (the largest
number of distinct dwords that fall into one of the 64 banks), for several orders of the work list and tile row strides.
Prints cycles per read instruction; 1.0 would be conflict free."""
import sys, numpy as np
sys.path[:0]=["/root/repo","/root/repo/tests"]
from openvslam_amd import synth
rng=np.random.default_rng(0)
img=synth.make_frame(1241,376,seed=5).astype(np.int32)
H,W=img.shape
thr=20
T=(thr+1)>>2
# ring offsets (dy,dx) as load_ring
S=None
ring=[(3,0),(3,1),(2,2),(1,3),(0,3),(-1,3),(-2,2),(-3,1),(-3,0),(-3,-1),(-2,-2),(-1,-3),(0,-3),(1,-3),(2,-2),(3,-1)]
pairs=[(0,8),(2,10),(4,12),(6,14)]
v6=img>>2
def shifted(a,dy,dx):
    out=np.zeros_like(a); 
    ys=slice(max(dy,0),H+min(dy,0)); yd=slice(max(-dy,0),H+min(-dy,0))
    xs=slice(max(dx,0),W+min(dx,0)); xd=slice(max(-dx,0),W+min(-dx,0))
    out[yd,xd]=a[ys,xs]; return out
R=[shifted(v6,dy,dx) for dy,dx in ring]
dark=np.ones((H,W),bool); bright=np.ones((H,W),bool)
for a,b in pairs:
    dark&=(R[a]<=v6-T)|(R[b]<=v6-T)
    bright&=(R[a]>=v6+T)|(R[b]>=v6+T)
surv=dark|bright
print("survivor fraction",surv[19:-19,19:-19].mean())
def cost(items_rc, stride_dw, offs=ring+[(0,0)]):
    # items_rc: array (n,2) of (row, col) within cell interior (0..63); tile byte = (r+3)*stride*4 + c+4
    n=len(items_rc); tot=0; ideal=0
    for g0 in range(0,n,64):
        it=items_rc[g0:g0+64]
        for dy,dx in offs:
            byte=(it[:,0]+3+dy)*stride_dw*4+it[:,1]+4+dx
            dw=byte>>2; bank=dw%64
            # cycles = max over banks of number of distinct dwords
            c=0
            for b in np.unique(bank):
                c=max(c,len(np.unique(dw[bank==b])))
            tot+=c; ideal+=1
    return tot,ideal
def cell_items(ci,cj,order):
    y0=19+ci*64+3; x0=19+cj*64+3
    s=surv[y0:y0+64,x0:x0+64]
    d=dark[y0:y0+64,x0:x0+64]
    if s.shape!=(64,64): return None
    items=[]
    if order=="current":
        # waves 0..3; lane = half<<5 | rowgroup<<3 | pair; r_first=2*wave+half+8*rowgroup; rows r_first, r_first+32
        for wave in range(4):
            for lane in range(64):
                pair=lane&7; r=2*wave+(lane>>5)+8*((lane>>3)&3)
                for k in range(2):
                    rr=r+32*k
                    for j in range(8):
                        c=8*pair+j
                        if s[rr,c]: items.append((rr,c,d[rr,c]))
    elif order=="ksplit":
        for wave in range(4):
            for k in range(2):
                for lane in range(64):
                    pair=lane&7; r=2*wave+(lane>>5)+8*((lane>>3)&3); rr=r+32*k
                    for j in range(8):
                        c=8*pair+j
                        if s[rr,c]: items.append((rr,c,d[rr,c]))
    elif order=="raster":
        for rr in range(64):
            for c in range(64):
                if s[rr,c]: items.append((rr,c,d[rr,c]))
    elif order=="colmajor":
        for c in range(64):
            for rr in range(64):
                if s[rr,c]: items.append((rr,c,d[rr,c]))
    items=np.array(items).reshape(-1,3)
    # dark first, then bright-only (reverse order from back): keep simple: stable partition
    dk=items[items[:,2]==1][:,:2]; br=items[items[:,2]==0][:,:2]
    return np.concatenate([dk,br[::-1]]) if len(items) else items[:,:2]
res={}
cells=[(ci,cj) for ci in range(0,5) for cj in range(0,18,2)]
for order in ("current","ksplit","raster","colmajor"):
    for stride in (18,17,19,20):
        tot=ideal=0; nitems=0
        for ci,cj in cells:
            it=cell_items(ci,cj,order)
            if it is None or len(it)==0: continue
            t,i=cost(it,stride); tot+=t; ideal+=i; nitems+=len(it)
        print(order,"stride",stride,"items",nitems,"cycles/instr %.2f"%(tot/ideal))
