#!/bin/bash
# One GPU visit of the BiLSTM-load reproducer: `gpurun -- bash tools/repro_visit.sh TAG [quick]`.  Writes gpurun_out/${TAG}_repro*.log:
# RAS / ECC counters of the box before and after, the victim x aggressor matrix in the default build, the narrow-conv rows in the
# two bisection builds and on split CU masks, and the library's own canary test three times.
TAG=${1:?tag}; MODE=${2:-full}
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ras() {
  echo "--- RAS / ECC ($1)"
  for d in /sys/class/drm/card*/device; do
    [ -d $d/ras ] || continue
    for f in $d/ras/*_err_count; do [ -r $f ] && echo "$f: $(tr '\n' ' ' < $f)"; done
  done
  rocm-smi --showrasinfo all 2>/dev/null | grep -v "^$" | head -60
}
L=$OUT/${TAG}_repro.log
{ ras before; } > $OUT/${TAG}_ras_before.log 2>&1
NARROW=xs_k3_n32,xs_k3_n64,xs_k7_n32,xs_k11_n32
{
  echo "=== default build: all victims x all aggressors"
  timeout 600 tools/bin/lstm_load_repro all all 200 3
  echo "=== B = 32 victims (the two-stream throughput schedule's BiLSTMs), N = 100"
  timeout 600 tools/bin/lstm_load_repro synth_single,synth_coop,real_coop none,xs_k3_n32,xs_k7_n32,xs_k7_n128 60 2 32 100
  if [ "$MODE" != quick ]; then
    echo "=== no s_setprio in the conv's k loop"
    timeout 300 tools/bin/lstm_load_repro_noprio synth_coop,real_coop,real_single none,$NARROW 200 3
    echo "=== predicated staging loads (no slot-0 re-reads)"
    timeout 300 tools/bin/lstm_load_repro_pred synth_coop,real_coop,real_single none,$NARROW 200 3
    echo "=== disjoint CU masks for the two streams"
    REPRO_MASK=split timeout 300 tools/bin/lstm_load_repro synth_coop,real_coop,real_single none,$NARROW 200 3
  fi
} > $L 2>&1
{ ras after; } > $OUT/${TAG}_ras_after.log 2>&1
diff $OUT/${TAG}_ras_before.log $OUT/${TAG}_ras_after.log > $OUT/${TAG}_ras_diff.log
for i in 1 2 3; do
  timeout 600 python -m pytest tests -q -m gpu -k "reproducible_next_to" -p no:cacheprovider 2>&1 | tail -4
done > $OUT/${TAG}_canary.log 2>&1
grep -c "bad_calls= *[1-9]" $L | sed "s/^/rows with bad calls: /"
grep "bad_calls= *[1-9]" $L | head -40
tail -5 $OUT/${TAG}_canary.log
