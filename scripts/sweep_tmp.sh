timeout 600 python -m pytest tests -m gpu -x -q -k "thread_per_query or config5 or align or ties" 2>&1 | tail -3
for st in 10 100; do
timeout 600 python bench.py --workload livox --steps $st --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); l = d['icp_last_launch']
print('steps $st: scans/s %.1f  ms/frame %.3f  us/iter %.1f  iters/frame %.1f | last launch: first %.1f us  later %.1f us  iters %d | frac %.3f' % (d['value'], d['ms_per_step'], 1000*d['ms_per_icp_iter'], d['config']['icp_iters_per_frame'], l['first_iteration_us'], l['later_iterations_us'], l['iterations'], d['roofline']['frac']))"
done
timeout 400 python scripts/icp_probe.py livox=1 frames=100 > gpurun_out/r04_m_icp_probe_livox100.txt 2>&1
