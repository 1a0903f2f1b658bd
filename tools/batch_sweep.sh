for bsz in 16 32 48 64 96 128; do python bench.py --batch $bsz --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/sw_$bsz.json; python - <<PY
import json
d=json.loads(open("gpurun_out/sw_$bsz.json").read().strip().splitlines()[-1])
print($bsz, round(d["ms_per_step"],3), round(d["value"]))
PY
done
