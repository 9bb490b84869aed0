import sqlite3, sys, re
db=sqlite3.connect(sys.argv[1]); c=db.cursor()
rows=list(c.execute("select name, start, end-start, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
idx=[i for i,r in enumerate(rows) if 'FusedAdam' in r[0]]
steps=len(idx)-1
a,b=idx[0]+1, idx[-1]+1
sel=rows[a:b]
tot=sum(r[2] for r in sel)
print("steps",steps,"kernel-busy ms/step",tot/steps/1e6,"launches/step",len(sel)/steps, "wall/step", (rows[b-1][1]-rows[a][1])/steps/1e6)
agg={}
for r in sel:
    n=re.sub(r'\(.*','',r[0])
    n=n.replace('void ','')
    if n.startswith('at::native'): n='torch:'+n[12:60]
    agg.setdefault(n,[0,0]); agg[n][0]+=r[2]; agg[n][1]+=1
tt=0;tc=0
for n,(t,cn) in sorted(agg.items(), key=lambda x:-x[1][0]):
    if n.startswith('torch:'): tt+=t; tc+=cn; continue
    print(f"{t/steps/1e3:8.1f} us {cn/steps:6.1f}  {n[:80]}")
print(f"{tt/steps/1e3:8.1f} us {tc/steps:6.1f}  all torch-native kernels")
