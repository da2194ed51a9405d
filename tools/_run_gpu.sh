timeout 1500 python -m pytest tests -m gpu -q -k "m2dp or config3 or config5 or fused or cluster" 2>&1 | grep -E "passed|failed"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); e = d['extra']; print(e['m2dp_match_50k']['queries_per_s'], e['m2dp_match_50k']['ms_per_step'], e['fused_1m_shard_fp16']['queries_per_s'], e['fused_1m_shard_fp16']['ms_per_step'])
"
timeout 600 python tools/fuzz_all.py 91 20 match,group,matcher,fused 2>&1 | grep -E "^BAD|fuzz_all:"
