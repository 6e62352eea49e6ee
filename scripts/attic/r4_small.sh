#!/bin/bash
# round 4: small clusters through the padded wavefront sweep, A/B against the previous routes (DPILQR_RICCATI_NO_PAD=1)
out=gpurun_out/r4b; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -x -k "six_state or three_state or twelve_state or padded or test_passes or solve_misc" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
rm -f $out/small.txt
for pad in 0 1; do
  if [ $pad == 1 ]; then export DPILQR_RICCATI_NO_PAD=1; else unset DPILQR_RICCATI_NO_PAD; fi
  echo "DPILQR_RICCATI_NO_PAD=$pad" >> $out/small.txt
  python scripts/bench_wg.py --model quad6 1 2 3 4 5 >> $out/small.txt 2>&1
  python scripts/bench_wg.py --model car3 1 2 3 4 5 6 >> $out/small.txt 2>&1
  python scripts/bench_wg.py --model quad12 --B 512 1 2 >> $out/small.txt 2>&1
  python scripts/solve_breakdown.py --model quad6 1 3 >> $out/small.txt 2>&1
done
cat $out/small.txt
