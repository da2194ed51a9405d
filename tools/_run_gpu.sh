cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tcp
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum -d /tmp/tcp -o sc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /tmp/tcp.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/tcp -name "*_results.db") | grep -E "sc_match_e" | cut -c1-200
tail -3 /tmp/tcp.log | cut -c1-200
