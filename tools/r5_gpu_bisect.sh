#!/bin/bash
# bisect of the bench.py segfault (round 5, session 2)
O=gpurun_out/s4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rook.py -x -q 2>&1 | tail -30 > $O/rook.log
for fl in "--no-floor" "--no-floor --no-components" ""; do
  n=x$(echo $fl | tr -d ' -')
  timeout 900 python3 -X faulthandler bench.py $fl > $O/b_$n.out 2> $O/b_$n.err; echo "rc=$? flags=$fl" >> $O/summary.txt
  tail -c 6000 $O/b_$n.err > $O/b_$n.err.tail; mv $O/b_$n.err.tail $O/b_$n.err
done
cat $O/summary.txt; tail -5 $O/rook.log
