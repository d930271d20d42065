timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py tests/test_scene_ext.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_cpp_host.py -x -q -m gpu -k "million or instanced or bvh or scene or lattice or forced or refit or fuzz or instance" 2>&1 | tail -4
for i in 1 2; do python bench.py --workload instanced --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; done
RTPT_NO_TRACE_FUSION=1 python bench.py --workload instanced --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
