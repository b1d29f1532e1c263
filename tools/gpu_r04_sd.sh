cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_sd; mkdir -p $O
( time timeout 900 python bench.py --steps 8 --repeat 1 --sds-steps 2 --posed-frames 0 --no-occupancy --no-cpu-baseline --sd-arch-steps 3 > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r04_sd/bench.json").read().strip().splitlines()[-1])
print(json.dumps(r.get("sds_step_sd_arch_standin"))[:1500])
PY
