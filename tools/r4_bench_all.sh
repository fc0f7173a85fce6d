#!/bin/bash
# Runs ON THE GPU BOX: the four configs' bench lines at the class defaults.
OUT=gpurun_out/r4_bench_all
mkdir -p $OUT
for c in imagenet language flow multimodal; do
  echo "== bench $c"; timeout -k 10 500 python bench.py --config $c --cpu-sample 0 > $OUT/$c.json 2> $OUT/$c.err || { tail -3 $OUT/$c.err | cut -c1-600; }
done
python - <<'PY'
import json
for c in ("imagenet","language","flow","multimodal"):
    try:
        d=json.load(open(f"gpurun_out/r4_bench_all/{c}.json"))
    except Exception as e:
        print(c,"ERR",e); continue
    print(c, "value", round(d["value"],1), "ms", round(d["ms_per_step"],3), "mfma", round(d["model_mfma_frac"],4), d["precision_policy"], "parity", f'{d["parity"]["relL2"]:.2e}/{d["parity"]["max_abs_over_absmax"]:.2e}', "eager", round(d.get("other_launch_mode",{}).get("ms_per_step",0),3))
    if "stages" in d: print("   stages", {k: round(v["ms"],4) for k,v in d["stages"].items()}, "roofline.frac", round(d["roofline"]["frac"],4))
    for k in ("class_default_policy","single_sweep_policy","plain_fp16_policy"):
        if k in d: print("   ",k, d[k]["policy"], round(d[k]["value"],1), d[k].get("parity"))
PY
