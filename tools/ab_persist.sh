for i in 1 2; do
for v in 0 1; do
HEXGNN_NO_PERSIST=$v python bench.py --config MIX --steps 50 --warmup 10 --no-cpu-baseline --no-split --sustain-s 0 --no-collective-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MIX no_persist=$v', d['value'], d['ms_per_step'])"
done; done
HEXGNN_NO_PERSIST=0 python bench.py --config MIX --eager --steps 50 --warmup 10 --no-cpu-baseline --no-split --sustain-s 0 --no-collective-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MIX eager persist', d['value'], d['ms_per_step'])"
