timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -12
for st in 10 100; do
timeout 600 python bench.py --workload livox --steps $st --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); l = d['icp_last_launch']
print('livox steps $st: scans/s %.1f  ms/frame %.3f  us/iter %.1f | last launch: first %.1f us  later %.1f us  iters %d | frac %.3f' % (d['value'], d['ms_per_step'], 1000*d['ms_per_icp_iter'], l['first_iteration_us'], l['later_iterations_us'], l['iterations'], d['roofline']['frac']))"
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('kitti 20/5: scans/s %.1f ms/frame %.4f us/iter %.2f' % (d['value'], d['ms_per_step'], 1000*d['ms_per_icp_iter']))"
timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('kitti 200/10: scans/s %.1f ms/frame %.4f us/iter %.2f' % (d['value'], d['ms_per_step'], 1000*d['ms_per_icp_iter']))"
timeout 600 python bench.py --workload mulran --steps 60 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('mulran: scans/s %.1f ms/frame %.4f us/iter %.2f' % (d['value'], d['ms_per_step'], 1000*d['ms_per_icp_iter']))"
