timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -12
for st in 100; do
timeout 600 python bench.py --workload livox --steps $st --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); l = d['icp_last_launch']
print('livox steps $st: scans/s %.1f  ms/frame %.3f  us/iter %.1f | last launch: first %.1f us  later %.1f us  iters %d | frac %.3f' % (d['value'], d['ms_per_step'], 1000*d['ms_per_icp_iter'], l['first_iteration_us'], l['later_iterations_us'], l['iterations'], d['roofline']['frac']))"
done
