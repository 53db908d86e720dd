"""Emulation of the class-tiled cell order of k_tower8_c128 (4 positions) with explicit swizzle keys (cz_conv_kernel.h): the
row map is a bijection, its inverse is the kernel's decode, which (tile, tap) pairs are off the board, and how many lanes of each
16-lane ds_read_b128 group share a slot (bank conflicts) per tile and tap.  A "group" here is 16 consecutive GEMM rows of a tile:
the kernel relabels its lanes (m31) so that the hardware's lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} own exactly those.  usage: python tools/experiments/trunk_layout_emulation.py"""
import itertools
def row_of(p,y,x):
    if y==0: return 8+8*p+x if x<8 else 2*p+(x-8)
    if x==0: return 40+8*p+(y-1)
    if x==9: return 168+8*p+(y-1)
    if y==8: return 72+8*p+(x-1)
    j=56*p+8*(y-1)+(x-1)
    return 104+j if j<64 else 136+j
def key_of(p,y,x):
    return 8*((y+p)&1) + ((x + y) & 7)
cells={}
for p in range(4):
    for y in range(9):
        for x in range(10):
            k=row_of(p,y,x); assert k not in cells, (p,y,x,k); cells[k]=(p,y,x)
assert sorted(cells)==list(range(360))
def cell_of_k(k):
    # the inverse, as the kernel would compute it
    if k<8: return (k//2,0,8+k%2)
    if k<40: return ((k-8)//8,0,(k-8)%8)
    if k<72: q=k-40; return (q//8, q%8+1, 0)
    if k<104: q=k-72; return (q//8, 8, q%8+1)
    if k<168: j=k-104
    elif k<200: q=k-168; return (q//8, q%8+1, 9)
    else: j=k-136
    p=j//56; r=j%56; return (p, r//8+1, r%8+1)
for k in range(360): assert cell_of_k(k)==cells[k], k
# tiles
conf_total=0; report={}
skippable={}
for tile in range(12):
    for tap in range(9):
        dy,dx=tap//3-1,tap%3-1
        allinv=True
        for grp in range(2):
            keys=[]
            for l in range(16):
                g=32*tile+16*grp+l; k=g-24
                if k<0:
                    keys.append(('pad',l&15)); continue
                p,y,x=cells[k]; yy,xx=y+dy,x+dx
                valid = 0<=yy<9 and 0<=xx<10
                if valid: allinv=False
                keys.append(key_of(p,yy,xx))
            ks=[kk if not isinstance(kk,tuple) else kk[1] for kk in keys]
            # pads: count separately (their key can be chosen freely) -> only real lanes
            real=[kk for kk in keys if not isinstance(kk,tuple)]
            c=len(real)-len(set(real))
            if c: report[(tile,tap,grp)]=c; conf_total+=c
        skippable[(tile,tap)]=allinv
print("conflicting lanes (real lanes only):", conf_total)
by_tile={}
for (tile,tap,grp),c in report.items(): by_tile[tile]=by_tile.get(tile,0)+c
print("by tile:", by_tile)
print("skippable tile-taps:", sorted(k for k,v in skippable.items() if v))
# write-side: own keys distinct per 16-lane group?
for tile in range(12):
    for grp in range(2):
        real=[]
        for l in range(16):
            k=32*tile+16*grp+l-24
            if k>=0: real.append(key_of(*cells[k]))
        if len(real)!=len(set(real)): print("own-key conflicts tile",tile,grp,len(real)-len(set(real)))
print({k:v for k,v in report.items() if k[0] in (2,6)})
# detail one
for (tile,tap,grp) in [(2,1,0),(2,4,0)]:
    dy,dx=tap//3-1,tap%3-1
    out=[]
    for l in range(16):
        k=32*tile+16*grp+l-24; p,y,x=cells[k]; out.append(((p,y,x),(p,y+dy,x+dx),key_of(p,y+dy,x+dx)))
    print(tile,tap,grp,out)
