for o in "icp_weight_long_emul=2" "icp_weight_long_emul=3" "icp_weight_long_emul=2 --opt icp_weight_long_base=192" "icp_weight_long_emul=4 --opt icp_weight_long_base=256"; do
  echo "== $o"; timeout 300 python bench.py --workload livox --steps 10 --warmup 3 --no-cpu-baseline --no-extras --opt $o 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); l = d['icp_last_launch']
print('scans/s %.1f  ms/frame %.3f  first %.1f us  later %.1f us  iters %d' % (d['value'], d['ms_per_step'], l['first_iteration_us'], l['later_iterations_us'], l['iterations']))"
done
echo "== probe emul=3"; timeout 300 python scripts/icp_probe.py livox=1 icp_weight_long_emul=3 | grep -A12 "per workgroup, iteration 6" | cut -c1-120
