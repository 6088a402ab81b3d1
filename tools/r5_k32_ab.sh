#!/bin/bash
# round 5: A/B on one box: config 4's triplet lines with the passes' 4-byte keys (A, the tree as it is) and without (B: the
# condition in sort_msd edited out on the box, library rebuilt there), twice each
run() {
ESP_EXTRA_ONLY=cfg4 ESP_CFG4_ONLY_TRIPLETS=1 ESP_BENCH_NO_DIGEST=1 timeout 900 python tools/r4_extra.py 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items():
        if 'triplets' in k: print('$1', k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('ms','key_bytes','stage_ms','error')})
"
}
run A1
cp extendablesparse.jl_amd/csrc/partition.hip /tmp/partition.hip.keep
sed -i 's/rem_out <= 32 \&\& rem_out <= esplocal::MAX_REM_BITS/false \&\& rem_out <= 32/' extendablesparse.jl_amd/csrc/partition.hip
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
run B1
cp /tmp/partition.hip.keep extendablesparse.jl_amd/csrc/partition.hip
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
run A2
sed -i 's/rem_out <= 32 \&\& rem_out <= esplocal::MAX_REM_BITS/false \&\& rem_out <= 32/' extendablesparse.jl_amd/csrc/partition.hip
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
run B2
