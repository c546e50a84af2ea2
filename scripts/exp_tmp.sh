for i in 1 2; do python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['config']['map_voxels'], d['config']['icp_workgroups'], d['config']['icp_iters_per_frame'], d['roofline']['ms_per_launch'], d['device_resident'], d['host_float32_input']['same_trajectory_as_host_input'])
"; done
