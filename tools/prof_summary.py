import sqlite3, sys, collections
c=sqlite3.connect(sys.argv[1]); steps=int(sys.argv[2])
d=collections.defaultdict(list)
for n,t in c.execute("select name,(end-start) from kernels"): d[n].append(t)
tot=sum(sum(v) for v in d.values())
for n,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:16]:
    print(f"{sum(v)/steps/1e6:8.3f} ms/step {len(v)//steps:5d} calls avg {sum(v)/len(v)/1e3:9.1f} us  {n[:100]}")
print("total/step", tot/steps/1e6)
