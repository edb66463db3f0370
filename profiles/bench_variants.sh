B="bench.py --steps 20 --warmup 5 --no-ncc --no-cpu-baseline"
for rep in 1 2; do
for v in "" "MI_X_DYN=1" "MI_X_DYN=1 MI_Z_DYN=1"; do
  echo "== $v" >> gpurun_out/r3_b2.txt
  env $v python3 $B 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print(d['ms_per_step'], r['pass_ms'])
" >> gpurun_out/r3_b2.txt
done
done
python3 profiles/overlap_probe.py > gpurun_out/r3_overlap2.txt 2>&1
