cd $GRAFT_REPO_ROOT
for wl in 4k 1080p v23-1080p; do
for cfg in ${SWEEP:-"4 2" "8 4" "12 4" "16 4" "8 8" "16 8" "8 4" "4 2" "4 4"}; do
  set -- $cfg
  python bench.py --workload $wl --streams $1 --cu-parts $2 --steps 120 --no-extra --no-cpu-baseline --no-live-traffic --no-configs --no-host-path --no-sustained 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['extra']['frames_per_s_repeated_regions']
print('$wl streams $1 parts $2: value %.1f median %.1f (p10 %.1f p90 %.1f)' % (d['value'], r['median'], r['p10'], r['p90']))"
done; done
