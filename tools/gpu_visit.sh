#!/bin/bash
# One GPU visit = `gpurun -- bash tools/gpu_visit.sh TAG STAGE [STAGE ...]`; every stage writes gpurun_out/${TAG}_*.
# (Rounds 1-3 kept one script per visit, 58 of them: they are in the history, `git log -- tools/`.)
#
#   probe            tools/probe_box.py FIRST: box fingerprint + the discriminating conv class in every build; a slow box
#                    (rule build of k7 / C256 / L8000 > 0.9 ms) runs `slowkit` on the spot
#   hunt             the quick probe only; a slow box then gets the full probe, the kit and both bench lines
#   slowkit          counters and ablations of that launch class: rocprofv3 --pmc passes (TCC hit / miss, FETCH_SIZE,
#                    busy cycles), tools/bin/xs_bench_{0,1,2,4,8,15} on the shape, per-workgroup timeline (xs_bench_64)
#   tests[:FILES]    pytest -m gpu (all, or the comma-separated test files)
#   smoke            __graft_entry__.smoke()
#   bench[:ARGS]     the driver's command `python bench.py --gpus 1 --steps 20 --warmup 5` (+ comma-separated extra args)
#   bench_ab         the same without the autotuner (--no-autotune), for the A/B on this box
#   configs          one short bench line per other configuration (libritts_hifigan, libritts_istftnet, longform)
#   stats[:SCHED]    rocprofv3 --kernel-trace --stats of a short bench on one schedule (default single): per-kernel table +
#                    tools/trace_gaps.py (busy / idle / overlapped time of the steady-state steps)
#   pmc              FETCH_SIZE / WRITE_SIZE / busy-cycle passes over tools/probe_dom.py (profiles/pmc_dominant.json)
#   headroom         tools/headroom_report.py on the synthetic / trained-like / small-magnitude checkpoints (both ends of the f16 range)
#   validate         tools/validate_checkpoint.py on the synthetic full-layout checkpoints
#   stats_cfg:NAME   rocprofv3 --kernel-trace --stats of a short single-stream bench of configuration NAME
#   pmc_narrow       FETCH_SIZE / WRITE_SIZE / MFMA-busy passes over tools/probe_narrow.py (fused conv, HiFi-GAN narrow stages)
#   smallgrid        tools/bin/xs_bench_0 on the B = 1 shapes: tile width 128 / 64 / 32 columns and the library's geometry rule
#   cmd:COMMAND      anything else (spaces as '+')
TAG=${1:?tag}; shift
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
run_slowkit() {
  for m in 0 1 2 4 8 15; do
    [ -x tools/bin/xs_bench_$m ] && timeout 120 tools/bin/xs_bench_$m 7 1 256 8000 32 1 1 10 2>&1 | tail -1 | tee -a $OUT/${TAG}_slowkit_xs_bench.log
  done
  for v in 0 1 2 3; do
    [ -x tools/bin/xs_bench_0 ] && XS_VARIANT=$v timeout 120 tools/bin/xs_bench_0 7 1 256 8000 32 1 1 10 2>&1 | tail -1 | sed "s/^/variant $v: /" | tee -a $OUT/${TAG}_slowkit_xs_bench.log
  done
  [ -x tools/bin/xs_bench_64 ] && timeout 120 tools/bin/xs_bench_64 7 1 256 8000 32 1 1 3 0 $OUT/${TAG}_slowkit_timeline.txt | tail -2
  for ctr in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    n=$(echo $ctr | tr ' ' '_')
    ( cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$n -o pmc -- python $OLDPWD/tools/probe_box.py --quick --level 0 --no-health > /dev/null 2>&1 )
    python tools/pmc_summary.py /tmp/pmc_$n 2>/dev/null | grep -i "conv1d_xs" | head -8 | sed "s/^/[$n] /" | tee -a $OUT/${TAG}_slowkit_pmc.txt
  done
}
for st in "$@"; do
  name=${st%%:*}; arg=""; [[ "$st" == *:* ]] && arg=${st#*:}
  echo "=== stage $st ($(date +%T))"
  case $name in
    probe)
      timeout 600 python tools/probe_box.py --out $OUT/${TAG}_box.json --level 1 2> $OUT/${TAG}_box.err | tail -1 | tee $OUT/${TAG}_box_class.txt
      grep -q "BOX_CLASS slow" $OUT/${TAG}_box_class.txt && { echo "SLOW BOX: running the kit"; run_slowkit; } ;;
    slowkit) run_slowkit ;;
    hunt_strong)  # is this a box of the strong class?  tools/bin/xs_bench_old_epilogue = the conv kernel as it was BEFORE the
      # row-end fix of the epilogue (whole-tile generic code at the end of a row): 128 x 256 tiles at k7 / C256 / L8000 took
      # 0.66 ms on most boxes and 1.07-1.11 ms on the strong class.  On such a box: the current library's probe and bench lines.
      old=$(XS_VARIANT=1 timeout 120 tools/bin/xs_bench_old_epilogue 7 1 256 8000 32 1 1 10 2>&1 | grep -o "[0-9.]* ms / launch" | cut -d" " -f1)
      echo "old epilogue, 128x256 tiles: $old ms" | tee $OUT/${TAG}_old_epilogue.txt
      if python -c "import sys; sys.exit(0 if float('${old:-0}') > 0.9 else 1)"; then
        echo "STRONG BOX"
        timeout 600 python tools/probe_box.py --level 0 --out $OUT/${TAG}_box.json 2> $OUT/${TAG}_box.err | tail -1
        bash tools/gpu_visit.sh $TAG bench:--no-cpu-baseline bench_ab
      fi ;;
    hunt)  # cheap: ~20 s on a fast box; on a slow one the whole kit + both bench lines + kernel statistics
      timeout 300 python tools/probe_box.py --quick --level 0 --out $OUT/${TAG}_box.json 2> $OUT/${TAG}_box.err | tail -1 | tee $OUT/${TAG}_box_class.txt
      if grep -q "BOX_CLASS slow" $OUT/${TAG}_box_class.txt; then
        echo "SLOW BOX: kit + bench lines"
        timeout 600 python tools/probe_box.py --level 1 --out $OUT/${TAG}_box_full.json 2>> $OUT/${TAG}_box.err | tail -1
        [ -n "${ST2_HUNT_KIT:-}" ] && run_slowkit
        bash tools/gpu_visit.sh $TAG bench:--no-cpu-baseline
        # the same box without the healthy-CU streams, and the long-form configuration both ways
        timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --cu-mask off --no-cpu-baseline --no-box-probe > $OUT/${TAG}_bench_nomask.json 2> $OUT/${TAG}_bench_nomask.err
        python tools/bench_summary.py $OUT/${TAG}_bench_nomask.json | head -8
        for m in auto off; do
          timeout 600 python bench.py --config longform --cu-mask $m --no-cpu-baseline --no-box-probe > $OUT/${TAG}_bench_longform_mask_$m.json 2> $OUT/${TAG}_bench_longform_mask_$m.err
          python tools/bench_summary.py $OUT/${TAG}_bench_longform_mask_$m.json | head -1
        done
      fi ;;
    tests)
      files="tests"; [ -n "$arg" ] && files=$(echo $arg | tr ',' ' ')
      timeout 2400 python -m pytest $files -m gpu -x -q 2>&1 | tail -15 | tee $OUT/${TAG}_pytest.log ;;
    smoke) timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/${TAG}_smoke.log ;;
    bench)
      extra=$(echo $arg | tr ',' ' ')
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 $extra > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
      tail -4 $OUT/${TAG}_bench.err; python tools/bench_summary.py $OUT/${TAG}_bench.json ;;
    bench_ab)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-autotune --no-cpu-baseline --no-box-probe > $OUT/${TAG}_bench_noautotune.json 2> $OUT/${TAG}_bench_noautotune.err
      python tools/bench_summary.py $OUT/${TAG}_bench_noautotune.json ;;
    configs)
      for c in libritts_hifigan libritts_istftnet longform; do
        timeout 600 python bench.py --config $c --no-cpu-baseline --no-box-probe > $OUT/${TAG}_bench_$c.json 2> $OUT/${TAG}_bench_$c.err
        python tools/bench_summary.py $OUT/${TAG}_bench_$c.json
      done ;;
    stats)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --schedule ${arg:-single} --calib-steps 0 --no-cpu-baseline --no-box-probe --other-configs off > $OLDPWD/$OUT/${TAG}_bench_single.json 2> $OLDPWD/$OUT/${TAG}_bench_single.err )
      f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_bench_single_kernel_stats.csv && head -25 $f
      t=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/trace_gaps.py $t | tee $OUT/${TAG}_trace_gaps.txt
      rm -rf /tmp/prof_$TAG ;;
    pmc)
      for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        n=$(echo $ctr | tr ' ' '_')
        ( cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcd_$n -o pmc -- python $OLDPWD/tools/probe_dom.py > /dev/null 2>&1 )
        python tools/pmc_summary.py /tmp/pmcd_$n 2>/dev/null | tee $OUT/${TAG}_pmc_$n.txt | head -12
      done
      python tools/pmc_summary.py --json $OUT/${TAG}_pmc_dominant.json --kernel "conv1d_xs_kernel" /tmp/pmcd_FETCH_SIZE /tmp/pmcd_WRITE_SIZE | tail -2 ;;
    pmc_narrow)  # the same three counter passes over tools/probe_narrow.py: the fused conv on the HiFi-GAN narrow stages (C = 64 / 32) and k = 3, C = 128
      for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        n=$(echo $ctr | tr ' ' '_')
        ( cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcn_$n -o pmc -- python $OLDPWD/tools/probe_narrow.py > $OLDPWD/$OUT/${TAG}_probe_narrow.log 2>&1 )
        python tools/pmc_summary.py /tmp/pmcn_$n 2>/dev/null | grep -i "f16s" | tee $OUT/${TAG}_pmc_narrow_$n.txt | head -12
      done ;;
    headroom)  # two-sided operand tables by rule / calibrated: synthetic, trained-like, small-magnitude checkpoints
      for v in "plain:" "trained:--trained-like" "small2:--trained-like+--small+1e-2" "small3:--trained-like+--small+1e-3" "libri_small2:--config+libritts+--trained-like+--small+1e-2"; do
        n=${v%%:*}; fl=$(echo ${v#*:} | tr '+' ' ')
        timeout 600 python tools/headroom_report.py $fl --json $OUT/${TAG}_headroom_$n.json > $OUT/${TAG}_headroom_$n.txt 2>&1
        echo "[$n] rc=$?"; grep -E "top of the range|calibration table|status word" $OUT/${TAG}_headroom_$n.txt
      done ;;
    validate)  # the first-contact tool on the synthetic full-layout checkpoints (no real checkpoint exists offline)
      for tag in ljspeech libritts; do
        python - <<PY
import sys, pathlib
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_checkpoint_layout as T
d = pathlib.Path("/tmp/ckpt_$tag"); d.mkdir(exist_ok=True)
print(*T._write_checkpoint(d, "$tag")[2:])
PY
        timeout 900 python tools/validate_checkpoint.py /tmp/ckpt_$tag/epoch_2nd_00017.pth /tmp/ckpt_$tag/config_libritts.yml --batch 2 --json $OUT/${TAG}_validate_$tag.json > $OUT/${TAG}_validate_$tag.txt 2>&1
        echo "[$tag] rc=$?"; tail -22 $OUT/${TAG}_validate_$tag.txt
      done ;;
    stats_cfg)  # rocprofv3 --kernel-trace --stats of a short single-stream bench of another configuration
      cfgname=${arg:-libritts_hifigan}
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$cfgname -o bench -- python $OLDPWD/bench.py --config $cfgname --steps 4 --warmup 2 --schedule single --calib-steps 0 --calibrate off --other-configs off --no-cpu-baseline --no-box-probe --cu-mask off > $OLDPWD/$OUT/${TAG}_bench_${cfgname}_single.json 2> $OLDPWD/$OUT/${TAG}_bench_${cfgname}_single.err )
      f=$(find /tmp/prof_${TAG}_$cfgname -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_${cfgname}_kernel_stats.csv && head -22 $f
      rm -rf /tmp/prof_${TAG}_$cfgname ;;
    smallgrid)  # tools/bin/xs_bench_0 at B = 1 (long-form / latency shapes): 128- / 64- / 32-column tiles and the geometry rule
      # arg = the batch sizes to sweep (default 1): how many 128 x 128 tiles until the narrow tiles stop paying?
      for bb in $(echo ${arg:-1} | tr ',' ' '); do
      for shp in "7 1 256 5680" "7 3 256 8000" "11 1 256 5680" "3 1 256 5680" "7 1 128 28400" "7 1 128 40000" "11 5 128 37200" "3 1 128 37200" "3 1 512 800" "3 1 1024 400"; do
        for v in 0 4 8 -1; do
          XS_VARIANT=$v timeout 120 tools/bin/xs_bench_0 $shp $bb 1 1 ${REPS:-50} 2>&1 | grep -E "ms / launch|checksum_y|tile columns|partial sums" | tr '\n' ' ' | sed "s/^/variant $v: /"; echo
        done
      done
      done | tee $OUT/${TAG}_smallgrid_xs_bench.log ;;
    cmd) timeout 1200 bash -c "$(echo $arg | tr '+' ' ')" 2>&1 | tail -300 | tee $OUT/${TAG}_cmd.log ;;
    *) echo "unknown stage $st" ;;
  esac
done
echo "=== visit $TAG done ($(date +%T))"
