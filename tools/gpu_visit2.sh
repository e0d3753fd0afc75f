#!/bin/bash
# Short GPU visit: selected tests, torchrun launch path (1 rank), LibriTTS/HiFi-GAN stage probe, bench + kernel trace.
set -u
TAG=${1:-r01j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest (pipeline + lstm + bert)"; timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "long_form or lstm or plbert or text_to_waveform or style_fc" > $OUT/pytest_sel.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest_sel.log
echo "== torchrun 1 rank"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_torchrun.json 2> $OUT/bench_torchrun.err; echo "torchrun exit $?"; cut -c1-300 $OUT/bench_torchrun.json; tail -3 $OUT/bench_torchrun.err
echo "== probe e2e libritts (HiFi-GAN, 10 steps)"; PROBE_TAG=libritts PROBE_STEPS=10 timeout 600 python tools/probe_e2e.py > $OUT/probe_e2e_libritts.log 2>&1; tail -4 $OUT/probe_e2e_libritts.log
echo "== rocprof trace"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/bench_prof.json 2> $R/$OUT/bench_prof.err ); echo "rocprof exit $?"
for f in $(find /tmp/prof_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/; done
for f in $(find /tmp/prof_$TAG -name '*kernel_trace.csv'); do head -1 $f > $OUT/kernel_trace_header.txt; python - "$f" "$OUT/kernel_trace_tail.csv.gz" <<'PY'
import csv, gzip, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-2600:]
with gzip.open(sys.argv[2], "wt") as f:
    w = csv.writer(f)
    w.writerow(["name", "start_ns", "dur_ns", "grid", "wg"])
    t0 = int(tail[0]["Start_Timestamp"])
    for r in tail:
        w.writerow([r["Kernel_Name"][:80], int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size", ""), r.get("Workgroup_Size", "")])
PY
done
ls -la $OUT
