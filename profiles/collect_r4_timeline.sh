#!/bin/bash
# Round-4 step timeline: per-dispatch start/end of every kernel of a short default bench run (both streams).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash profiles/collect_r4_timeline.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_timeline
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-f32-line --no-gather-bench --no-extra-configs"
python3 $B --steps 50 --warmup 10 > $O/bench_n1.json 2> $O/bench_n1.err
NTF_BENCH_MIN_TIMED_S=0.01 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $B --steps 6 --warmup 3 > $O/trace.log 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null
ls -la $O $O/trace/* | head -30
