#!/bin/bash
# kernel dispatches per training step: difference of two rocprofv3 kernel traces with different step counts (initialisation cancels)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
root=$PWD
cd /tmp
for k in 4 14; do
  rm -rf /tmp/kt$k
  rocprofv3 --kernel-trace --stats -d /tmp/kt$k -o kt -- python $root/bench.py --steps $k --warmup 2 --no-cpu-baseline --no-profile --no-other-configs > /tmp/kt$k.log 2>&1
  python $root/tools/rocpd_summary.py $(find /tmp/kt$k -name "*.db" | head -1) 400 > /tmp/kt$k.md
  grep "total kernel time" /tmp/kt$k.md
done
python - <<'PY'
import re
def load(p):
    d={}
    for l in open(p):
        m=re.match(r"\| `(.+?)` \| (\d+) \| ([\d.]+) \|", l)
        if m: d[m.group(1)]=(int(m.group(2)), float(m.group(3)))
    return d
a,b=load('/tmp/kt4.md'),load('/tmp/kt14.md')
tot=0; rows=[]
for k,(n,t) in b.items():
    n0,t0=a.get(k,(0,0.0))
    dn=(n-n0)/10.0; dt=(t-t0)/10.0
    tot+=dn; rows.append((dn,dt,k))
print('launches per step: %.1f' % tot)
rows.sort(reverse=True)
for dn,dt,k in rows[:40]:
    print('%7.1f  %8.3f ms  %s' % (dn, dt, k[:90]))
PY
