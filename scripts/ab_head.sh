# A/B of the one-kernel inference head against the torch layers, same box, back to back.
# Prints ms/step and the stage table of every run; --per-step adds the per-step series and
# a log of Python garbage collections (the reason a mean over 32 steps can mislead).
set -e
R=${GRAFT_REPO_ROOT:-.}
line() { grep '^{"metric"' $1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['config']['head'], '|', d['config']['host'], '|', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stages'].items()})"; }
for f in "" "--no-fused-head" "" "--no-fused-head"; do
  timeout -k 10 200 python -u $R/bench.py --no-cpu-baseline $f > $R/gpurun_out/ab.log 2>&1; line $R/gpurun_out/ab.log "[$f]"
done
for f in "--no-gc-freeze" "--no-gc-freeze --no-fused-head"; do
  timeout -k 10 200 python -u $R/bench.py --no-cpu-baseline $f > $R/gpurun_out/ab.log 2>&1; line $R/gpurun_out/ab.log "[$f]"
done
