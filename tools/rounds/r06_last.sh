#!/bin/bash
# round 6, the last word on the final tree: suite, smoke, default bench line, randomized sweeps on NEW seeds (one gpurun call)
set -u
O=gpurun_out/r06p; mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "suite: $(tail -1 $O/pytest.log)" | tee $O/summary.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke: $(tail -1 $O/smoke.log)" | tee -a $O/summary.txt
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-330 | tee -a $O/summary.txt
{ echo "== consistency (WaveformModes) seeds 300..499"; timeout 900 python tools/consistency_sweep.py 500 300 2>&1 | tail -2
  echo "== consistency (AsymptoticBondiData) seeds 150..249"; timeout 900 env SWEEP_ABD=1 python tools/consistency_sweep.py 250 150 2>&1 | tail -2
  echo "== series calculus seeds 1500..2499"; timeout 600 python tools/series_sweep.py 2500 1500 2>&1 | tail -2
  echo "== fuzz seeds 400..699"; timeout 900 python tools/fuzz_sweep.py 700 400 2>&1 | tail -1; } > $O/sweeps.txt 2>&1
cat $O/sweeps.txt
