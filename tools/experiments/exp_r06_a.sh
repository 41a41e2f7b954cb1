#!/bin/bash
# round 6, call A: which arrangement shows the 2-rank BC-gradient deviation, and which tensor differs first (tools/diag_r06.py)
O=gpurun_out/r06_a; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for m in "poison 1 2" "solo 2 3" "pg 8 3" "pgraw 8 3" "nopg 8 3" "lock 6 3" "hammer 4 3"; do
  set -- $m
  timeout 900 python tools/diag_r06.py $1 $2 $3 > $O/$1.log 2>&1
  echo "== $1 rc $?"; grep -c "differ" $O/$1.log
done
VPT_POISON=big timeout 300 python tools/diag_r06.py poison 1 2 > $O/poison_big.log 2>&1
timeout 900 python tools/diag_r06.py pg 6 3 AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 > $O/pg_serialize.log 2>&1
tail -n 40 $O/pg.log
