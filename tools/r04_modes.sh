#!/bin/bash
# round 4: the mono pipeline alternates between a smooth (187 k) and a bumpy (172 k, a 400-us delivery gap every ~10 steps)
# mode inside one run -- which knob moves it?  Step traces of short variants.
set -o pipefail
O=gpurun_out/modes
mkdir -p $O
WL=kitti00_mono_1241x376_n1000
run() { name=$1; shift; timeout -k 10 240 env "$@" python bench.py --workload $WL --inputs device --no-cpu-baseline --min-seconds 1.5 --stamp-dump $O/st_$name ${EXTRA[@]} > $O/b_$name.json 2> $O/b_$name.err; echo "$name rc=$? $(python3 -c "import json; d=json.load(open('$O/b_$name.json')); print(d['value'], d.get('spread'))" 2>/dev/null)"; }
EXTRA=(); run default A=1
EXTRA=(--fast-chain); run chain A=1
EXTRA=(--fast-chain); run chain2 A=1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/modes/st_*.json')):
    d=json.load(open(f))
    w=[e for e in d['events'] if e[0]=='w']
    deliv=[x[3] for x in w]
    K=100
    per=32 if 'b24' not in f else 24
    rates=[per*K/(deliv[(r+1)*K-1]-deliv[r*K-1])/1e3 for r in range(1,len(deliv)//K)]
    print(f.split('st_')[1].split('.')[0], ' '.join('%.0f'%x for x in rates))
PY
echo done
