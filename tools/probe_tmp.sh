set -e
cd /root/repo
export TMPDIR=/tmp
python3 tools/interactive_rate.py --calls 300 2>&1 | grep -v '^W\|^E\|amdgpu.ids'
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['value']), j['ms_per_step'], {k:round(v/j['steps'],2) for k,v in j['roofline']['stage_ms'].items()})"
