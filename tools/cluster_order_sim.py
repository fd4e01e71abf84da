#!/usr/bin/env python3
"""CPU simulation of the pair search's culls (cluster test, then the lane test against cluster A's bounding box, area
range and centres) on the benchmark's candidate lists, for other orders of the list than the anchor order: how many
boxes of B would still be broadcast against A's 64 lanes.  (Result: ~5 400 per image in every order tried -- column
strips of 64..192 px, Morton order of the cells, area class + Morton order of the centres -- so reordering the list
buys nothing.)"""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/pytorch-retinanet_amd')
import bench
from oracle import dense_head_oracle as O
size=608; C=80
anc = O.anchors_for_image(size,size)[0]
n=anc.shape[0]
icls, ireg = bench.synth_infer(np.random.default_rng(3000), 2, n, C)
def decode(anc, reg):
    w=anc[:,2]-anc[:,0]; h=anc[:,3]-anc[:,1]; cx=anc[:,0]+0.5*w; cy=anc[:,1]+0.5*h
    dx=reg[:,0]*0.1; dy=reg[:,1]*0.1; dw=reg[:,2]*0.2; dh=reg[:,3]*0.2
    pcx=cx+dx*w; pcy=cy+dy*h; pw=np.exp(dw)*w; ph=np.exp(dh)*h
    b=np.stack([pcx-0.5*pw,pcy-0.5*ph,pcx+0.5*pw,pcy+0.5*ph],1)
    b[:,0]=np.maximum(b[:,0],0); b[:,1]=np.maximum(b[:,1],0); b[:,2]=np.minimum(b[:,2],size); b[:,3]=np.minimum(b[:,3],size)
    return b
# level geometry
shapes=[(size+2**l-1)//2**l for l in (3,4,5,6,7)]
off=np.cumsum([0]+[s*s*9 for s in shapes])
def survivors(order, boxes):
    b=boxes[order]; K=len(b); nc=(K+63)//64
    area=(b[:,2]-b[:,0])*(b[:,3]-b[:,1]); cx=(b[:,0]+b[:,2])/2; cy=(b[:,1]+b[:,3])/2
    cl=[slice(i*64,min(K,(i+1)*64)) for i in range(nc)]
    bb=np.array([[b[s,0].min(),b[s,1].min(),b[s,2].max(),b[s,3].max()] for s in cl])
    ar=np.array([[area[s].min(),area[s].max()] for s in cl])
    cb=np.array([[cx[s].min(),cy[s].min(),cx[s].max(),cy[s].max()] for s in cl])
    f=0.5*0.9999
    pairs=0; surv=0
    for A in range(nc):
        for B in range(A,nc):
            if bb[B,2]<=bb[A,0] or bb[B,0]>=bb[A,2] or bb[B,3]<=bb[A,1] or bb[B,1]>=bb[A,3]: continue
            if ar[B,0]*f>ar[A,1] or ar[B,1]<ar[A,0]*f: continue
            # centre rule
            if cb[B,2]<bb[A,0] or cb[B,0]>bb[A,2] or cb[B,3]<bb[A,1] or cb[B,1]>bb[A,3]: continue
            if cb[A,2]<bb[B,0] or cb[A,0]>bb[B,2] or cb[A,3]<bb[B,1] or cb[A,1]>bb[B,3]: continue
            pairs+=1
            s=cl[B]
            ok=(b[s,2]>bb[A,0])&(b[s,0]<bb[A,2])&(b[s,3]>bb[A,1])&(b[s,1]<bb[A,3])
            ok&=(area[s]*f<=ar[A,1])&(area[s]>=ar[A,0]*f)
            ok&=(cx[s]>=bb[A,0])&(cx[s]<=bb[A,2])&(cy[s]>=bb[A,1])&(cy[s]<=bb[A,3])
            surv+=int(ok.sum())
    return pairs,surv
for j in range(2):
    sc=icls[j].max(1); cand=np.nonzero(sc>0.05)[0]
    boxes=decode(anc[cand], ireg[j][cand])
    base=np.arange(len(cand))
    print("image",j,"K",len(cand),"anchor order:",survivors(base,boxes))
    # alternative orders: by (level, x block of W cells, y, x)
    lvl=np.searchsorted(off,cand,side='right')-1
    r=cand-off[lvl]; cell=r//9; s=np.array(shapes)[lvl]; y=cell//s; x=cell%s
    for blk_px in (64,96,128,192):
        xb=(x*(2**(lvl+3)))//blk_px
        key=np.lexsort((r%9,x,y,xb,lvl))
        print("   column strips of",blk_px,"px:",survivors(key,boxes))
    # morton within level
    def part1by1(v):
        v=v.astype(np.uint64); v=(v|(v<<8))&0x00FF00FF; v=(v|(v<<4))&0x0F0F0F0F; v=(v|(v<<2))&0x33333333; v=(v|(v<<1))&0x55555555; return v
    mort=part1by1(x)|(part1by1(y)<<1)
    key=np.lexsort((r%9,mort,lvl))
    print("   morton:",survivors(key,boxes))
    # by size class then morton of centre
    area=(boxes[:,2]-boxes[:,0])*(boxes[:,3]-boxes[:,1]); cls_=np.floor(np.log2(np.maximum(area,1))).astype(int)
    cxq=((boxes[:,0]+boxes[:,2])/2/8).astype(int); cyq=((boxes[:,1]+boxes[:,3])/2/8).astype(int)
    mort2=part1by1(cxq)|(part1by1(cyq)<<1)
    key=np.lexsort((mort2,cls_))
    print("   area class + morton of centre:",survivors(key,boxes))


def survivors_sub(order, boxes, sub):
    """As survivors(), with cluster A cut into `sub` runs of consecutive entries: a box of B is kept only if
    it passes the lane test against at least one run's bounding box / area range / centres."""
    b = boxes[order]; K = len(b); nc = (K + 63) // 64
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]); cx = (b[:, 0] + b[:, 2]) / 2; cy = (b[:, 1] + b[:, 3]) / 2
    cl = [slice(i * 64, min(K, (i + 1) * 64)) for i in range(nc)]
    bb = np.array([[b[s, 0].min(), b[s, 1].min(), b[s, 2].max(), b[s, 3].max()] for s in cl])
    ar = np.array([[area[s].min(), area[s].max()] for s in cl])
    cb = np.array([[cx[s].min(), cy[s].min(), cx[s].max(), cy[s].max()] for s in cl])
    f = 0.5 * 0.9999
    step = 64 // sub
    surv = 0
    for A in range(nc):
        runs = [slice(A * 64 + q * step, min(K, A * 64 + (q + 1) * step)) for q in range(sub)]
        runs = [r for r in runs if r.start < K]
        rb = [(b[r, 0].min(), b[r, 1].min(), b[r, 2].max(), b[r, 3].max(), area[r].min(), area[r].max()) for r in runs]
        for B in range(A, nc):
            if bb[B, 2] <= bb[A, 0] or bb[B, 0] >= bb[A, 2] or bb[B, 3] <= bb[A, 1] or bb[B, 1] >= bb[A, 3]: continue
            if ar[B, 0] * f > ar[A, 1] or ar[B, 1] < ar[A, 0] * f: continue
            if cb[B, 2] < bb[A, 0] or cb[B, 0] > bb[A, 2] or cb[B, 3] < bb[A, 1] or cb[B, 1] > bb[A, 3]: continue
            if cb[A, 2] < bb[B, 0] or cb[A, 0] > bb[B, 2] or cb[A, 3] < bb[B, 1] or cb[A, 1] > bb[B, 3]: continue
            s = cl[B]
            ok = np.zeros(s.stop - s.start, bool)
            for (x1, y1, x2, y2, amin, amax) in rb:
                o = (cx[s] >= x1) & (cx[s] <= x2) & (cy[s] >= y1) & (cy[s] <= y2)
                o &= (area[s] * f <= amax) & (area[s] >= amin * f)
                ok |= o
            surv += int(ok.sum())
    return surv


for j in range(2):
    sc = icls[j].max(1); cand = np.nonzero(sc > 0.05)[0]
    boxes = decode(anc[cand], ireg[j][cand])
    base = np.arange(len(cand))
    print("image", j, "runs of A:", {sub: survivors_sub(base, boxes, sub) for sub in (1, 2, 4, 8, 16)})
