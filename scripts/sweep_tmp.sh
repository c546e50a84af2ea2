timeout 600 python -m pytest tests -m gpu -x -q -k "thread_per_query or config5 or align or ties or weight" 2>&1 | tail -3
for o in "icp_wide=-1" ; do
for st in 10 100; do
timeout 600 python bench.py --workload livox --steps $st --warmup 3 --no-cpu-baseline --no-extras --opt $o 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); l = d['icp_last_launch']
print('$o steps $st: scans/s %.1f  ms/frame %.3f  us/iter %.1f | last launch: first %.1f us  later %.1f us  iters %d | frac %.3f' % (d['value'], d['ms_per_step'], 1000*d['ms_per_icp_iter'], l['first_iteration_us'], l['later_iterations_us'], l['iterations'], d['roofline']['frac']))"
done; done
timeout 400 python scripts/icp_probe.py livox=1 frames=100 > gpurun_out/r04_t_icp_probe_livox100.txt 2>&1
