import itertools, sys
from run_search import run
g=sys.argv[1] if len(sys.argv)>1 else "1"; rv=sys.argv[2] if len(sys.argv)>2 else "1"
best=[]
for lw,lu,rw,ru,rt in itertools.product(range(3),range(3),range(2),range(2),range(2)):
    r=run(["-DGEBD2V="+g,"-DROTV="+rv,"-DSROTV="+rv,"-DLW=%d"%lw,"-DLU=%d"%lu,"-DRW=%d"%rw,"-DRU=%d"%ru,"-DRT=%d"%rt],"full")
    best.append((r[1]+r[2],r,(lw,lu,rw,ru,rt)))
best.sort(reverse=True)
for b in best[:5]: print(b)
