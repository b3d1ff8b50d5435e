for q in 2 4 8; do for fx in 0 1; do echo "== GPU_MAX_HW_QUEUES=$q FORCE_EXCHANGE=$fx"; GPU_MAX_HW_QUEUES=$q LSQ_BENCH_FORCE_EXCHANGE=$fx python bench.py --no-cpu --repeats 15 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['value_min'], d['value_max'], d['config'].get('exchange_backend'))"; done; done
