import sys
rows=[]
for ln in open(sys.argv[1]):
    if ln.startswith('#'): continue
    p=ln.split(None,7)
    rows.append((float(p[0]),float(p[1]),float(p[2]),p[7][:60].strip() if len(p)>7 else ''))
gaps=[(rows[i+1][0]-rows[i][1],i) for i in range(len(rows)-1)]
big=[g for g in gaps if g[0]>150]
idx=[0]+[g[1]+1 for g in big]+[len(rows)]
for a,b in zip(idx[:-1],idx[1:]):
    sel=rows[a:b]
    if len(sel)<50: continue
    span=sel[-1][1]-sel[0][0]; k=sum(r[2] for r in sel)
    print(f"--- section {a}:{b}: {len(sel)} dispatches span {span:.0f} kernel {k:.0f} gap {span-k:.0f}")
    if '--detail' in sys.argv:
        for i in range(len(sel)):
            g=sel[i][0]-sel[i-1][1] if i else 0
            print(f"{sel[i][0]-sel[0][0]:8.1f} +{g:6.1f} {sel[i][2]:7.1f} {sel[i][3]}")
