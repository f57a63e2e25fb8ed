#!/bin/bash
# Board power and clocks while the headline kernel runs back to back (read-only rocm-smi queries).
set -o pipefail
python3 -u tools/kbench.py --f16 --reps 5000 > gpurun_out/power_kbench.log 2>&1 &
PID=$!
sleep 12
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | grep -v "=====" | sed 's/^/   /'
  sleep 1
done
wait $PID
grep median gpurun_out/power_kbench.log
